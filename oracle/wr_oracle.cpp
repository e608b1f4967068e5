// wr_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the per-frame render path of nullprop/wgpu-renderer
// (src/core/state.rs:630-809 driving res/shaders/*.wgsl), written as a plain scalar
// immediate-mode software pipeline: every draw is rasterised in order, every fragment that
// passes the depth test is shaded and blended at once, exactly as the reference's wgpu
// passes would.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; it is the checker, never the thing measured or shipped.
//
// PARITY UNPINNED: the reference has no tests, golden images or fixtures for this path and
// cannot be built here (Rust + wgpu + a Vulkan driver, none present), so nothing pins this
// restatement to the reference's pixels.  The shader bodies follow the WGSL line by line
// (cited below); everything the reference leaves to wgpu/naga/the Vulkan driver is fixed
// here by an explicit rule (list: "driver-defined rules").
//
// Arithmetic: f32, every operation rounded separately in source order (build with
// -ffp-contract=off); fmaf only where written.  Matrices are column-major m[col][row].
//
// Driver-defined rules (each one is a decision of this restatement):
//  R1  mat*vec  = ((c0*v0 + c1*v1) + c2*v2) + c3*v3 ; dot3 = (x*x' + y*y') + z*z'.
//  R2  normalize(v) = v * (1/sqrt(dot(v,v))) ; length = sqrt(dot).
//  R3  mix(a,b,t) = a*(1-t) + b*t ; smoothstep(0,1,v): t = clamp(v,0,1), t*t*(3-2t);
//      fract(x) = x - floor(x) ; saturate = min(max(x,0),1).
//  R4  sin = single-precision Cephes sinf as sse_mathfun's sin_ps has it and Mesa's gallivm
//      (lp_build_sin_or_cos, the lowering lavapipe runs for SPIR-V Sin) follows it: octant
//      j = (trunc(|x|*4/pi) + 1) & ~1, r = |x| - j*pi/4 with pi/4 in three parts
//      (0.78515625, 2.4187564849853515625e-4, 3.77489497744594108e-8), degree-3 polynomials
//      in z = r*r for sine (j&2 == 0) and cosine, sign = sign(x) ^ (j&4); fused multiply-adds
//      in the reduction and the Horner steps; |x| >= 2^30 or NaN gives +0.  Mesa's source is
//      not in this image: the constants and the step order are restated from the published
//      Cephes / sse_mathfun code, unpinned against a lavapipe run.  A rule of this list
//      changes only on evidence from the reference or from the specifications it runs on,
//      never for speed.
//  R5  pow(b, 5.0) = (b*b)*(b*b)*b.
//  R6  clip volume: -w<=x,y<=w, 0<=z<=w.  A triangle with all vertices outside one plane
//      is dropped.  A triangle with any vertex with z<0, z>w, |x|>8w or |y|>8w is clipped
//      (Sutherland-Hodgman; planes in the order z>=0, z<=w, x>=-8w, x<=8w, y>=-8w, y<=8w;
//      a new vertex is in + t*(out-in), t = d_in/(d_in-d_out)) and fan-triangulated
//      (p0,p_i,p_i+1); the fan index is the "sub" index (0..6).
//  R7  viewport: x_w = (x/w)*(W/2) + W/2, y_w = (y/w)*(-H/2) + H/2 (Y flipped as wgpu does
//      on Vulkan), z_w = z/w with 1/w computed once; vertices snap to 1/256 pixel,
//      X = floor(x_w*256 + 0.5).
//  R8  front face = counter-clockwise in y-up NDC (negative area in the y-down window);
//      back faces and zero-area triangles are dropped (cull_mode Back everywhere).
//  R9  coverage: sample at pixel centres, exact integer edge functions, top-left rule.
//  R10 lambda_i = float(E_i) * (1/float(2*area)); z = (z0 + l1*(z1-z0)) + l2*(z2-z0),
//      clamped to [0,1]; perspective: p_i = l_i/w_i, q_i = p_i * (1/((p0+p1)+p2));
//      clipped triangles map q through the clip weights to the original vertices;
//      a varying = (B0*a0 + B1*a1) + B2*a2.
//  R11 depth bias (float depth, Vulkan rule): o = 2.0*max(|dz/dx|,|dz/dy|) + 2*2^(e-23),
//      e = exponent of the largest vertex z of the (sub-)triangle; biased z clamped to [0,1].
//  R12 derivatives: per 2x2 quad, ddx = v(x0+1,y0)-v(x0,y0), ddy = v(x0,y0+1)-v(x0,y0),
//      (x0,y0) the quad's even-aligned top-left pixel, evaluated on the fragment's own
//      triangle (extrapolated where the quad mate is not covered).
//  R13 material sampler (Repeat, mag Linear, min Nearest, one level):
//      rho = max(max(|dudx|,|dudy|)*W, max(|dvdx|,|dvdy|)*H); rho <= 1 -> bilinear
//      (x = u*W-0.5, texels wrap, lerp a + f*(b-a), x first then y, in f32 on decoded
//      texels), else nearest floor(u*W).  sRGB textures decode per texel before filtering.
//  R14 shadow sampler (ClampToEdge, Linear, compare LessEqual): four compares at
//      x = u*S-0.5, weights as R13, texel indices clamped.  Depth texture sampling in the
//      fog pass is plain bilinear with clamped indices.
//  R15 colour target Rgba8UnormSrgb: blend in linear f32 on the decoded 8-bit destination,
//      out = src*sa + dst*(1-sa) for rgb AND alpha (pass.rs:67-83); encode rgb to the
//      nearest sRGB code (ties decided on exact double thresholds), alpha = floor(a*255+0.5);
//      NaN encodes as 0.
//  R16 proj*view is one mat4*mat4 product (columns by R1) done once per frame.
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off -mfma)
//
// Alternatives of the rules that no second source checks (R1 inside rand, R3, R4, R12, R13's
// tie, R15's rounding) can be compiled in, one macro each (WRO_ALT_*), to measure how far a
// driver that decides differently would move the picture.  scripts/rule_sensitivity.py builds
// those variants into oracle/_variants/ and records the distances; libwr_oracle.so itself is
// always built without any of them, and no rule changes because of what they show.

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

#include "../include/wr_abi.h"

namespace {

// ------------------------------------------------------------------------- small math
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { float c[4][4]; };  // c[col][row]
struct M3 { float c[3][3]; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   // R1
inline float length3(V3 a) { return sqrtf(dot3(a, a)); }                       // R2
inline V3 normalize3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
#ifdef WRO_ALT_R3_MIX_ONE_MUL
inline float mixf(float a, float b, float t) { return a + t * (b - a); }          // R3 alternative: the flrp lowering
#else
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }  // R3
#endif
inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

inline V4 mul(const M4& m, V4 v) {  // R1
    V4 r;
    r.x = ((m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z) + m.c[3][0] * v.w;
    r.y = ((m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z) + m.c[3][1] * v.w;
    r.z = ((m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z) + m.c[3][2] * v.w;
    r.w = ((m.c[0][3] * v.x + m.c[1][3] * v.y) + m.c[2][3] * v.z) + m.c[3][3] * v.w;
    return r;
}
inline M4 mul(const M4& a, const M4& b) {  // R16
    M4 r;
    for (int j = 0; j < 4; ++j) {
        V4 col = mul(a, V4{b.c[j][0], b.c[j][1], b.c[j][2], b.c[j][3]});
        r.c[j][0] = col.x; r.c[j][1] = col.y; r.c[j][2] = col.z; r.c[j][3] = col.w;
    }
    return r;
}
inline V3 mul(const M3& m, V3 v) {
    V3 r;
    r.x = (m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z;
    r.y = (m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z;
    r.z = (m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z;
    return r;
}
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// R4: single-precision Cephes sine (the scheme of sse_mathfun's sin_ps, which Mesa's gallivm
// lowers fsin to), fused multiply-adds in the reduction and the Horner steps.
#ifdef WRO_ALT_R4_UNFUSED_SINE
// R4 alternative: the same scheme with every multiply and add rounded on its own (sse_mathfun
// as written, without the fused forms a compiler may pick).
inline float o_fma(float a, float b, float c) { return a * b + c; }
#else
inline float o_fma(float a, float b, float c) { return fmaf(a, b, c); }
#endif
float o_sin(float x) {
    uint32_t sign = f2u(x) & 0x80000000u;
    float ax = fabsf(x);
    if (!(ax < 1073741824.0f)) return 0.0f;
    float y = ax * 1.27323954473516f;  // 4/pi
    int j = (int)y;
    j = (j + 1) & ~1;
    y = (float)j;
    sign ^= ((uint32_t)(j & 4)) << 29;
    bool sin_poly = (j & 2) == 0;
    float r = o_fma(y, -0.78515625f, ax);
    r = o_fma(y, -2.4187564849853515625e-4f, r);
    r = o_fma(y, -3.77489497744594108e-8f, r);
    float z = r * r;
    float res;
    if (sin_poly) {
        float p = -1.9515295891e-4f;
        p = o_fma(p, z, 8.3321608736e-3f);
        p = o_fma(p, z, -1.6666654611e-1f);
        p = p * z;
        res = o_fma(p, r, r);
    } else {
        float p = 2.443315711809948e-5f;
        p = o_fma(p, z, -1.388731625493765e-3f);
        p = o_fma(p, z, 4.166664568298827e-2f);
        p = p * z;
        p = p * z;
        p = o_fma(z, -0.5f, p);
        res = p + 1.0f;
    }
    return u2f(f2u(res) ^ sign);
}

// ------------------------------------------------------------------------- noise.wgsl
struct Counters {
    uint64_t fog_fragments = 0, fog_noise_evals = 0, shaded_fragments = 0, blended_lt1 = 0;
    uint64_t fog_noise_evals_unused = 0;  // of fog_noise_evals: light marches of fog points in shadow (result never read)
};

// noise.wgsl:3-5
float o_hash_of_dot(float d) {
    float s = o_sin(d) * 42123.45f;
    float fr = s - floorf(s);
    return fr * 2.0f - 1.0f;
}
#ifdef WRO_ALT_R1_FUSED_DOT
// R1 alternative inside rand: the dot product as a chain of fused multiply-adds
float o_rand(V3 p) { return o_hash_of_dot(fmaf(p.z, 412.12f, fmaf(p.x, 12.345f, p.y * 67.89f))); }
#else
float o_rand(V3 p) { return o_hash_of_dot(dot3(p, v3(12.345f, 67.89f, 412.12f))); }
#endif
inline float smooth01(float v) {  // R3, smoothstep(0,1,v)
    float t = fminf(fmaxf(v, 0.0f), 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
// noise.wgsl:7-24
float o_value_noise(V3 p) {
    V3 u = v3(floorf(p.x), floorf(p.y), floorf(p.z));
    V3 v = p - u;
    V3 s = v3(smooth01(v.x), smooth01(v.y), smooth01(v.z));
    float a = o_rand(u);
    float b = o_rand(u + v3(1, 0, 0));
    float c = o_rand(u + v3(0, 1, 0));
    float d = o_rand(u + v3(1, 1, 0));
    float e = o_rand(u + v3(0, 0, 1));
    float f = o_rand(u + v3(1, 0, 1));
    float g = o_rand(u + v3(0, 1, 1));
    float h = o_rand(u + v3(1, 1, 1));
    return mixf(mixf(mixf(a, b, s.x), mixf(c, d, s.x), s.y),
                mixf(mixf(e, f, s.x), mixf(g, h, s.x), s.y), s.z);
}
// noise.wgsl:26-40
float o_fbm(V3 p) {
    float weight = 0.5f;
    V3 q = p;
    float ret = 0.0f;
    for (int i = 0; i < 8; ++i) {
        ret += weight * o_value_noise(q);
        q = q * 2.0f;
        weight *= 0.5f;
    }
    return saturatef(ret);
}
// fog.wgsl:48-62
float o_fog_noise(V3 pos, float time) {
    V3 p1 = pos * 0.01f;
    p1.x += time * 0.2f;
    p1.y += time * 0.2f;
    p1.z += o_sin(time * 0.1f) * 0.5f;
    float n1 = o_fbm(p1);
    V3 p2 = pos * 0.05f;
    p2.x += time * 0.2f;
    p2.y += time * 0.2f;
    p2.z += o_sin(time * 0.1f) * 0.5f;
    float n2 = o_fbm(p2);
    return (0.8f * n1 + 0.15f * n2) + 0.05f;
}

// ------------------------------------------------------------------------- sRGB (R15)
struct SrgbTables {
    float decode[256];
    float unorm[256];
    float thresh[256];  // thresh[k], k>=1: smallest linear value that encodes to >= k
    SrgbTables() {
        for (int c = 0; c < 256; ++c) {
            decode[c] = (float)to_linear((double)c / 255.0);
            unorm[c] = (float)c / 255.0f;
#ifdef WRO_ALT_R15_TRUNCATE
            thresh[c] = (float)to_linear((double)c / 255.0);  // R15 alternative: codes round down
#else
            thresh[c] = c == 0 ? 0.0f : (float)to_linear(((double)c - 0.5) / 255.0);
#endif
        }
    }
    static double to_linear(double s) {
        return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4);
    }
};
const SrgbTables g_srgb;

uint8_t srgb_encode(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
    int lo = 0, hi = 255;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (x >= g_srgb.thresh[mid]) lo = mid; else hi = mid - 1;
    }
    return (uint8_t)lo;
}
uint8_t unorm_encode(float a) {
    if (!(a > 0.0f)) return 0;
    if (a >= 1.0f) return 255;
#ifdef WRO_ALT_R15_TRUNCATE
    return (uint8_t)(int)(a * 255.0f);
#else
    return (uint8_t)(int)(a * 255.0f + 0.5f);
#endif
}

// ------------------------------------------------------------------------- scene
struct Texture { uint32_t w, h; int srgb; std::vector<uint8_t> px; };
struct Material { uint32_t base, normal, mr; float metallic, roughness; };
struct Model {
    std::vector<WrModelVertex> verts;
    std::vector<uint32_t> idx;       // global vertex indices (mesh base added)
    std::vector<uint32_t> tri_mat;   // material per triangle
    std::vector<Material> mats;
    WrInstanceRaw inst;
    bool has_inst = false;
};

struct Varyings {  // pbr.wgsl VertexOutput, globals.wgsl:55-62
    V4 clip;
    V2 uv;
    V3 tpos, tlight, tview;
    V4 wpos;
};

struct SubTri {
    int32_t X[3], Y[3];
    int64_t A[3], B[3], C[3];
    bool tl[3];
    int64_t area2;
    float z0, dz1, dz2, inv2A, invw[3];
    float beta[3][3];
    bool clipped;
    uint32_t tri, sub;
    int pxmin, pymin, pxmax, pymax;
    float bias;
};

struct ClipVert { float x, y, z, w, b[3]; };

const float GUARD = 8.0f;

inline float plane_dist(const ClipVert& v, int plane) {
    switch (plane) {
        case 0: return v.z;
        case 1: return v.w - v.z;
        case 2: return v.x + GUARD * v.w;
        case 3: return GUARD * v.w - v.x;
        case 4: return v.y + GUARD * v.w;
        default: return GUARD * v.w - v.y;
    }
}
inline ClipVert clip_lerp(const ClipVert& in, const ClipVert& out, float t) {
    ClipVert r;
    r.x = in.x + t * (out.x - in.x);
    r.y = in.y + t * (out.y - in.y);
    r.z = in.z + t * (out.z - in.z);
    r.w = in.w + t * (out.w - in.w);
    for (int k = 0; k < 3; ++k) r.b[k] = in.b[k] + t * (out.b[k] - in.b[k]);
    return r;
}

// R6: returns the number of polygon vertices (0 if nothing is left).
int clip_polygon(ClipVert* poly, int n) {
    ClipVert tmp[12];
    for (int plane = 0; plane < 6; ++plane) {
        bool any_out = false;
        for (int i = 0; i < n; ++i) if (!(plane_dist(poly[i], plane) >= 0.0f)) any_out = true;
        if (!any_out) continue;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const ClipVert& a = poly[i];
            const ClipVert& b = poly[(i + 1) % n];
            float da = plane_dist(a, plane), db = plane_dist(b, plane);
            bool ina = da >= 0.0f, inb = db >= 0.0f;
            if (ina) tmp[m++] = a;
            if (ina != inb) {
                if (ina) tmp[m++] = clip_lerp(a, b, da / (da - db));
                else     tmp[m++] = clip_lerp(b, a, db / (db - da));
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
        if (n < 3) return 0;
    }
    return n;
}

struct Target { int W, H; };

// R7-R11: set up one (sub-)triangle from three clip-space vertices.  Returns false when
// it is culled or covers no pixel centre.
bool setup_subtri(const ClipVert cv[3], Target tg, bool shadow, SubTri& st) {
    float sx = 0.5f * (float)tg.W, tx = 0.5f * (float)tg.W;
    float sy = -0.5f * (float)tg.H, ty = 0.5f * (float)tg.H;
    int32_t X[3], Y[3];
    float zw[3], iw[3];
    for (int k = 0; k < 3; ++k) {
        if (!(cv[k].w > 0.0f)) return false;
        float rw = 1.0f / cv[k].w;
        float xw = (cv[k].x * rw) * sx + tx;
        float yw = (cv[k].y * rw) * sy + ty;
        zw[k] = cv[k].z * rw;
        iw[k] = rw;
        float fx = floorf(xw * 256.0f + 0.5f), fy = floorf(yw * 256.0f + 0.5f);
        if (!(fabsf(fx) < 1.0e9f) || !(fabsf(fy) < 1.0e9f)) return false;
        X[k] = (int32_t)fx;
        Y[k] = (int32_t)fy;
    }
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) -
                    (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (area2 >= 0) return false;  // R8: back-facing or degenerate
    // reorder (0,2,1) so that the area is positive
    const int ord[3] = {0, 2, 1};
    float z[3];
    for (int k = 0; k < 3; ++k) {
        int s = ord[k];
        st.X[k] = X[s]; st.Y[k] = Y[s];
        z[k] = zw[s];
        st.invw[k] = iw[s];
        for (int j = 0; j < 3; ++j) st.beta[k][j] = cv[s].b[j];
    }
    st.area2 = -area2;
    // edge i is opposite vertex i: a = P[(i+1)%3], b = P[(i+2)%3]
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, b = (i + 2) % 3;
        st.A[i] = (int64_t)st.Y[a] - (int64_t)st.Y[b];
        st.B[i] = (int64_t)st.X[b] - (int64_t)st.X[a];
        st.C[i] = (int64_t)st.X[a] * (int64_t)st.Y[b] - (int64_t)st.X[b] * (int64_t)st.Y[a];
        st.tl[i] = (st.A[i] > 0) || (st.A[i] == 0 && st.B[i] > 0);
    }
    int xmin = std::min(st.X[0], std::min(st.X[1], st.X[2]));
    int xmax = std::max(st.X[0], std::max(st.X[1], st.X[2]));
    int ymin = std::min(st.Y[0], std::min(st.Y[1], st.Y[2]));
    int ymax = std::max(st.Y[0], std::max(st.Y[1], st.Y[2]));
    st.pxmin = std::max((xmin - 128 + 255) >> 8, 0);
    st.pymin = std::max((ymin - 128 + 255) >> 8, 0);
    st.pxmax = std::min((xmax - 128) >> 8, tg.W - 1);
    st.pymax = std::min((ymax - 128) >> 8, tg.H - 1);
    if (st.pxmin > st.pxmax || st.pymin > st.pymax) return false;
    st.inv2A = 1.0f / (float)st.area2;
    st.z0 = z[0];
    st.dz1 = z[1] - z[0];
    st.dz2 = z[2] - z[0];
    st.bias = 0.0f;
    if (shadow) {  // R11
        float k = 256.0f * st.inv2A;
        float dzdx = ((float)st.A[1] * st.dz1 + (float)st.A[2] * st.dz2) * k;
        float dzdy = ((float)st.B[1] * st.dz1 + (float)st.B[2] * st.dz2) * k;
        float m = fmaxf(fabsf(dzdx), fabsf(dzdy));
        float zmax = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], 0.0f));
        uint32_t e8 = (f2u(zmax) >> 23) & 0xffu;
        float r = e8 >= 24u ? u2f((e8 - 23u) << 23) : 0.0f;
        st.bias = m * 2.0f + r * 2.0f;
    }
    return true;
}

// R6: one input triangle -> sub-triangles (appended in fan order).
void setup_triangle(const V4 c[3], uint32_t tri, Target tg, bool shadow, std::vector<SubTri>& out) {
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(c[k].x) || !std::isfinite(c[k].y) || !std::isfinite(c[k].z) ||
            !std::isfinite(c[k].w)) return;
    bool rej[6] = {true, true, true, true, true, true};
    bool need_clip = false;
    for (int k = 0; k < 3; ++k) {
        const V4& v = c[k];
        if (!(v.x < -v.w)) rej[0] = false;
        if (!(v.x > v.w)) rej[1] = false;
        if (!(v.y < -v.w)) rej[2] = false;
        if (!(v.y > v.w)) rej[3] = false;
        if (!(v.z < 0.0f)) rej[4] = false;
        if (!(v.z > v.w)) rej[5] = false;
        float g = GUARD * v.w;
        if (v.z < 0.0f || v.z > v.w || v.x < -g || v.x > g || v.y < -g || v.y > g) need_clip = true;
    }
    if (rej[0] || rej[1] || rej[2] || rej[3] || rej[4] || rej[5]) return;
    ClipVert poly[12];
    for (int k = 0; k < 3; ++k) {
        poly[k].x = c[k].x; poly[k].y = c[k].y; poly[k].z = c[k].z; poly[k].w = c[k].w;
        poly[k].b[0] = k == 0; poly[k].b[1] = k == 1; poly[k].b[2] = k == 2;
    }
    int n = 3;
    if (need_clip) {
        n = clip_polygon(poly, 3);
        if (n < 3) return;
    }
    for (int i = 1; i + 1 < n; ++i) {
        ClipVert cv[3] = {poly[0], poly[i], poly[i + 1]};
        SubTri st;
        if (!setup_subtri(cv, tg, shadow, st)) continue;
        st.clipped = need_clip;
        st.tri = tri;
        st.sub = (uint32_t)(i - 1);
        out.push_back(st);
    }
}

// R9
inline bool covers(const SubTri& st, int px, int py, int64_t E[3]) {
    int64_t sx = (int64_t)px * 256 + 128, sy = (int64_t)py * 256 + 128;
    bool in = true;
    for (int i = 0; i < 3; ++i) {
        E[i] = st.A[i] * sx + st.B[i] * sy + st.C[i];
        if (!(E[i] > 0 || (E[i] == 0 && st.tl[i]))) in = false;
    }
    return in;
}
inline void edge_values(const SubTri& st, int px, int py, int64_t E[3]) {
    int64_t sx = (int64_t)px * 256 + 128, sy = (int64_t)py * 256 + 128;
    for (int i = 0; i < 3; ++i) E[i] = st.A[i] * sx + st.B[i] * sy + st.C[i];
}
// R10
inline float frag_depth(const SubTri& st, const int64_t E[3]) {
    float l1 = (float)E[1] * st.inv2A, l2 = (float)E[2] * st.inv2A;
    float z = (st.z0 + l1 * st.dz1) + l2 * st.dz2;
    return fminf(fmaxf(z, 0.0f), 1.0f);
}
inline void frag_bary(const SubTri& st, const int64_t E[3], float B[3]) {
    float l1 = (float)E[1] * st.inv2A, l2 = (float)E[2] * st.inv2A;
    float l0 = (1.0f - l1) - l2;
    float p0 = l0 * st.invw[0], p1 = l1 * st.invw[1], p2 = l2 * st.invw[2];
    float inv = 1.0f / ((p0 + p1) + p2);
    float q0 = p0 * inv, q1 = p1 * inv, q2 = p2 * inv;
    for (int j = 0; j < 3; ++j) B[j] = (q0 * st.beta[0][j] + q1 * st.beta[1][j]) + q2 * st.beta[2][j];
}
inline float ip(const float B[3], float a0, float a1, float a2) { return (B[0] * a0 + B[1] * a1) + B[2] * a2; }

// ------------------------------------------------------------------------- texture sampling
inline int wrapi(int i, int n) { i %= n; return i < 0 ? i + n : i; }
inline int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
inline float guard_coord(float x) { return (fabsf(x) < 1.0e9f) ? x : 0.0f; }

inline V4 texel(const Texture& t, int x, int y) {
    const uint8_t* p = &t.px[((size_t)y * t.w + x) * 4];
    if (t.srgb) return V4{g_srgb.decode[p[0]], g_srgb.decode[p[1]], g_srgb.decode[p[2]], g_srgb.unorm[p[3]]};
    return V4{g_srgb.unorm[p[0]], g_srgb.unorm[p[1]], g_srgb.unorm[p[2]], g_srgb.unorm[p[3]]};
}
inline float lerpf(float a, float b, float f) { return a + f * (b - a); }
inline V4 lerp4(V4 a, V4 b, float f) { return V4{lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f)}; }

// R13
V4 sample_material(const Texture& t, V2 uv, V2 ddx, V2 ddy) {
    float W = (float)t.w, H = (float)t.h;
    float rho = fmaxf(fmaxf(fabsf(ddx.x), fabsf(ddy.x)) * W, fmaxf(fabsf(ddx.y), fabsf(ddy.y)) * H);
#ifdef WRO_ALT_R13_TIE_MINIFIES
    if (rho < 1.0f) {  // R13 alternative: lambda = 0 takes the minification filter
#else
    if (rho <= 1.0f) {
#endif
        float x = guard_coord(uv.x * W - 0.5f), y = guard_coord(uv.y * H - 0.5f);
        float x0 = floorf(x), y0 = floorf(y);
        float fx = x - x0, fy = y - y0;
        int i0 = wrapi((int)x0, (int)t.w), j0 = wrapi((int)y0, (int)t.h);
        int i1 = wrapi(i0 + 1, (int)t.w), j1 = wrapi(j0 + 1, (int)t.h);
        V4 top = lerp4(texel(t, i0, j0), texel(t, i1, j0), fx);
        V4 bot = lerp4(texel(t, i0, j1), texel(t, i1, j1), fx);
        return lerp4(top, bot, fy);
    }
    float x = guard_coord(floorf(uv.x * W)), y = guard_coord(floorf(uv.y * H));
    return texel(t, wrapi((int)x, (int)t.w), wrapi((int)y, (int)t.h));
}

// ------------------------------------------------------------------------- frame state
struct Frame {
    int W, H, S;
    M4 view, proj, pv;
    M4 light_m[6];
    V3 cam_pos, light_pos;
    V4 light_color;
    V4 planes;
    float time;
    uint32_t use_shadowmaps;
    const float* shadow;  // 6*S*S
};

// R14 / light.wgsl:17-23 textureSampleCompare, Linear + LessEqual + ClampToEdge
inline float shadow_compare(const Frame& F, float u, float v, int layer, float ref) {
    float S = (float)F.S;
    float x = guard_coord(u * S - 0.5f), y = guard_coord(v * S - 0.5f);
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int i0 = clampi((int)x0, F.S), i1 = clampi((int)x0 + 1, F.S);
    int j0 = clampi((int)y0, F.S), j1 = clampi((int)y0 + 1, F.S);
    const float* L = F.shadow + (size_t)layer * F.S * F.S;
    float c00 = ref <= L[(size_t)j0 * F.S + i0] ? 1.0f : 0.0f;
    float c10 = ref <= L[(size_t)j0 * F.S + i1] ? 1.0f : 0.0f;
    float c01 = ref <= L[(size_t)j1 * F.S + i0] ? 1.0f : 0.0f;
    float c11 = ref <= L[(size_t)j1 * F.S + i1] ? 1.0f : 0.0f;
    return lerpf(lerpf(c00, c10, fx), lerpf(c01, c11, fx), fy);
}

// constants.wgsl:1-17 under the shaders' own names (PI is PI_C here: <cmath> may define PI-like
// macros).  INV_SHADOW_SAMPLES is 1/25 although the loops take 16 taps, as in the reference.
const float PI_C = 3.14159f;
const float INV_SQRT_2 = 0.70710678118654752440f;
const float INV_SQRT_3 = 0.57735026918962576451f;
const int SHADOW_SAMPLES = 2;
const float INV_SHADOW_SAMPLES = 1.0f / 25.0f;
const int FOG_MAX_STEPS = 30;
const float FOG_STEP_SIZE = 5.0f;
const int FOG_LIGHT_MAX_STEPS = 10;
const float FOG_LIGHT_STEP_SIZE = 10.0f;
const float FOG_DENSITY = 2.0f;
const float FOG_LIGHT_DENSITY = 3.0f;
const float FOG_ALPHA = 1.0f;
const float FOG_DENSITY_COLOR[2] = {0.5f, 0.1f};

// light.wgsl:1-29
float sample_direct_light_index(const Frame& F, int index, V4 lc) {
    if (lc.w <= 0.0f) return 0.0f;
    float proj_correction = 1.0f / lc.w;
    float lx = (lc.x * 0.5f) * proj_correction + 0.5f;
    float ly = (lc.y * -0.5f) * proj_correction + 0.5f;
    float ref = lc.z * proj_correction - 0.000001f;
    float total = 0.0f;
    float texel_size = (float)F.S;
    for (int x = -SHADOW_SAMPLES; x < SHADOW_SAMPLES; ++x)
        for (int y = -SHADOW_SAMPLES; y < SHADOW_SAMPLES; ++y) {
            float ox = (float)x / texel_size, oy = (float)y / texel_size;
            float s = shadow_compare(F, lx + ox, ly + oy, index, ref);
            total += s * INV_SHADOW_SAMPLES;
        }
    return total;
}

// light.wgsl:31-61
float sample_direct_light(const Frame& F, V4 wp) {
    float in_light = 0.0f;
    if (F.use_shadowmaps > 0u) {
        for (int i = 0; i < 6; ++i) {
            V4 lc = mul(F.light_m[i], wp);
            V3 ld = normalize3(v3(lc.x, lc.y, lc.z));
            if (ld.z < INV_SQRT_3 - 0.01f) continue;
            if (fabsf(ld.y) > INV_SQRT_2 + 0.01f) continue;
            if (fabsf(ld.x) > INV_SQRT_2 + 0.01f) continue;
            in_light = sample_direct_light_index(F, i, lc);
            if (in_light > 0.0f) break;
        }
    } else {
        in_light = 1.0f;
    }
    return in_light;
}

// light.wgsl:63-78
V3 sample_ambient_light(V4 light, float light_dist, float surface_light_dot) {
    V3 ambient = v3(0.01f, 0.01f, 0.01f);
    float att = 1.0f / ((1.0f + -0.75f * light_dist) + (0.25f * light_dist) * light_dist);
    V3 diff_direct = (v3(light.x, light.y, light.z) * light.w) * att;
    float diffuse_mult = fmaxf(surface_light_dot, 0.0f);
    return ambient + diff_direct * (diffuse_mult * 0.03f);
}

// brdf.wgsl:35-65
V3 brdf(V3 n, V3 l, V3 v, V3 h, V3 albedo, float roughness, float metalness) {
    V3 dialect = v3(mixf(0.04f, albedo.x, metalness), mixf(0.04f, albedo.y, metalness), mixf(0.04f, albedo.z, metalness));
    float cos_theta = fmaxf(dot3(h, v), 0.0f);
    float b = 1.0f - cos_theta;
    float b2 = b * b;
    float p5 = (b2 * b2) * b;  // R5
    V3 fresnel = dialect + (v3(1, 1, 1) - dialect) * p5;
    // distribution_ggx, brdf.wgsl:3-12
    float a2 = roughness * roughness;
    float ndh = fmaxf(dot3(n, h), 0.0f);
    float ndh2 = ndh * ndh;
    float den = ndh2 * (a2 - 1.0f) + 1.0f;
    den = (PI_C * den) * den;
    float ndf = a2 / den;
    // geometry_smith, brdf.wgsl:16-27
    float ndv = fmaxf(dot3(n, v), 0.0f);
    float ndl = fmaxf(dot3(n, l), 0.0f);
    float k = roughness;
    float ggx1 = ndv / (ndv * (1.0f - k) + k);
    float ggx2 = ndl / (ndl * (1.0f - k) + k);
    float geo = ggx1 * ggx2;
    V3 nom = fresnel * (ndf * geo);
    float denom = (4.0f * ndv) * ndl + 0.0001f;
    V3 specular = nom / denom;
    V3 kd = (v3(1, 1, 1) - fresnel) * (1.0f - metalness);
    return (((kd * albedo) / PI_C) + specular) * ndl;
}

struct Scene {
    Frame F;
    std::vector<Texture> tex;
    Model models[3];
};

// pbr.wgsl:73-130
V4 fs_pbr(const Scene& sc, const Material& m, V2 uv, V2 ddx, V2 ddy, V3 tpos, V3 tlight, V3 tview, V4 wpos) {
    const Frame& F = sc.F;
    V4 td = sample_material(sc.tex[m.base], uv, ddx, ddy);
    V4 tn = sample_material(sc.tex[m.normal], uv, ddx, ddy);
    V4 trm = sample_material(sc.tex[m.mr], uv, ddx, ddy);
    V3 albedo = v3(td.x, td.y, td.z);
    float roughness = trm.y * m.roughness;
    float metalness = trm.z * m.metallic;
    V3 total = v3(0, 0, 0);
    V3 normal_dir = v3(tn.x * 2.0f - 1.0f, tn.y * 2.0f - 1.0f, tn.z * 2.0f - 1.0f);
    V3 light_dir = normalize3(tlight - tpos);
    float sld = dot3(normal_dir, light_dir);
    float light_dist = length3(F.light_pos - v3(wpos.x, wpos.y, wpos.z));
    float att = 1.0f / (1.0f + light_dist * light_dist);
    V3 lrgb = v3(F.light_color.x, F.light_color.y, F.light_color.z);
    V3 direct = (lrgb * F.light_color.w) * att;
    float in_light = sample_direct_light(F, wpos);
    if (in_light > 0.0f) {
        V3 view_dir = normalize3(tview - tpos);
        V3 half_dir = normalize3(view_dir + light_dir);
        float rs = fmaxf(sld, 0.0f);
        V3 radiance = (direct * rs) * in_light;
        total = total + radiance * brdf(normal_dir, light_dir, view_dir, half_dir, albedo, roughness, metalness);
    }
    V3 ambient = sample_ambient_light(F.light_color, light_dist, sld);
    ambient = ambient * albedo;
    V3 result = ambient + total;
    result = result / (result + v3(1, 1, 1));
    return V4{result.x, result.y, result.z, td.w};
}

// fog.wgsl:118-123
inline float depth_to_linear(float depth, float zn, float zf) { return (zn * zf) / (zf + depth * (zn - zf)); }

// fog.wgsl:64-85
V2 ray_march(const Frame& F, V3 origin, V3 dir, float max_depth, int max_steps, float step_size, float fog_density, Counters& cnt) {
    float density = 0.0f, depth = 0.0f;
    for (int i = 0; i < max_steps; ++i) {
        depth += step_size;
        if (depth >= max_depth) break;
        float noise = o_fog_noise(origin + dir * depth, F.time);
        cnt.fog_noise_evals++;
        float contribution = fog_density / (float)max_steps;
        density += noise * contribution;
        if (density >= 1.0f) { density = 1.0f; break; }
    }
    return V2{density, depth};
}

// fog.wgsl:125-166 (+ ray_march_fog 87-116)
V4 fs_fog(const Scene& sc, const float* depth_tex, V3 origin, float frag_z, int px, int py, Counters& cnt) {
    const Frame& F = sc.F;
    V3 direction = normalize3(origin - F.cam_pos);
    float volume_depth = depth_to_linear(frag_z, F.planes.x, F.planes.y);
    float u = ((float)px + 0.5f) / F.planes.z, v = ((float)py + 0.5f) / F.planes.w;
    // textureSample(t_geometry_depth, s_geometry_depth, uv): R14 plain bilinear
    float gd;
    {
        float x = guard_coord(u * (float)F.W - 0.5f), y = guard_coord(v * (float)F.H - 0.5f);
        float x0 = floorf(x), y0 = floorf(y);
        float fx = x - x0, fy = y - y0;
        int i0 = clampi((int)x0, F.W), i1 = clampi((int)x0 + 1, F.W);
        int j0 = clampi((int)y0, F.H), j1 = clampi((int)y0 + 1, F.H);
        float t00 = depth_tex[(size_t)j0 * F.W + i0], t10 = depth_tex[(size_t)j0 * F.W + i1];
        float t01 = depth_tex[(size_t)j1 * F.W + i0], t11 = depth_tex[(size_t)j1 * F.W + i1];
        gd = lerpf(lerpf(t00, t10, fx), lerpf(t01, t11, fx), fy);
    }
    float geometry_depth = depth_to_linear(gd, F.planes.x, F.planes.y);
    float max_fog_depth = geometry_depth - volume_depth;
    if (max_fog_depth <= 0.0f) return V4{0, 0, 0, 0};
    // ray_march_fog
    V2 fm = ray_march(F, origin, direction, max_fog_depth, FOG_MAX_STEPS, FOG_STEP_SIZE, FOG_DENSITY, cnt);
    float fog_density = fm.x, fog_depth = fm.y;
    V3 fog_end = origin + direction * fog_depth;
    V3 to_light = F.light_pos - fog_end;
    float max_light_dist = length3(to_light);
    V3 light_direction = to_light / max_light_dist;
    const uint64_t evals_before_light_march = cnt.fog_noise_evals;
    V2 lm = ray_march(F, fog_end, light_direction, max_light_dist, FOG_LIGHT_MAX_STEPS, FOG_LIGHT_STEP_SIZE, FOG_LIGHT_DENSITY, cnt);
    const uint64_t light_march_evals = cnt.fog_noise_evals - evals_before_light_march;
    float occlusion = lm.x;
    V3 fog_position = origin + direction * fog_depth;
    float light_dist = length3(F.light_pos - fog_position);
    float bc = mixf(FOG_DENSITY_COLOR[0], FOG_DENSITY_COLOR[1], fog_density);
    V3 ambient = sample_ambient_light(F.light_color, light_dist, 1.0f) * 2.0f;
    ambient = ambient * v3(bc, bc, bc);
    V3 radiance = v3(0, 0, 0);
    float in_light = sample_direct_light(F, V4{fog_position.x, fog_position.y, fog_position.z, 1.0f});
    if (in_light > 0.0f) {
        float att = 1.0f / (1.0f + light_dist * light_dist);
        V3 lrgb = v3(F.light_color.x, F.light_color.y, F.light_color.z);
        radiance = (((lrgb * F.light_color.w) * att) * in_light) * (1.0f - occlusion);
    } else {
        cnt.fog_noise_evals_unused += light_march_evals;
    }
    V3 result = ambient + radiance;
    result = result / (result + v3(1, 1, 1));
    return V4{result.x, result.y, result.z, fog_density * FOG_ALPHA};
}

// R15
inline void blend_store(uint8_t* dst, V4 src, bool blend) {
    V4 o = src;
    if (blend) {
        float sa = src.w, ia = 1.0f - src.w;
        o.x = src.x * sa + g_srgb.decode[dst[0]] * ia;
        o.y = src.y * sa + g_srgb.decode[dst[1]] * ia;
        o.z = src.z * sa + g_srgb.decode[dst[2]] * ia;
        o.w = src.w * sa + g_srgb.unorm[dst[3]] * ia;
    }
    dst[0] = srgb_encode(o.x); dst[1] = srgb_encode(o.y); dst[2] = srgb_encode(o.z); dst[3] = unorm_encode(o.w);
}

M4 to_m4(const float m[4][4]) { M4 r; memcpy(r.c, m, 64); return r; }

struct Oracle {
    Scene sc;
    int W, H, S;
    std::vector<uint8_t> color;
    std::vector<float> depth;
    std::vector<uint32_t> vis;
    std::vector<float> shadow;
    Counters cnt;
    uint64_t n_subtris_geom = 0, n_subtris_shadow = 0;
};

template <class Fn>
void parallel_bands(int H, int threads, Fn fn) {
    threads = std::max(1, std::min(threads, H));
    if (threads == 1) { fn(0, 0, H); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) {
        int y0 = (int)((int64_t)H * t / threads), y1 = (int)((int64_t)H * (t + 1) / threads);
        th.emplace_back([=]() { fn(t, y0, y1); });
    }
    for (auto& t : th) t.join();
}

void render(Oracle& o, const WrCameraUniform* cam, const WrLightUniform* light, const WrGlobalUniforms* glob, uint32_t passes, int threads) {
    Scene& sc = o.sc;
    Frame& F = sc.F;
    F.W = o.W; F.H = o.H; F.S = o.S;
    F.view = to_m4(cam->view); F.proj = to_m4(cam->proj);
    F.pv = mul(F.proj, F.view);
    for (int i = 0; i < 6; ++i) F.light_m[i] = to_m4(light->matrices[i]);
    F.cam_pos = v3(cam->position[0], cam->position[1], cam->position[2]);
    F.light_pos = v3(light->position[0], light->position[1], light->position[2]);
    F.light_color = V4{light->color[0], light->color[1], light->color[2], light->color[3]};
    F.planes = V4{cam->planes[0], cam->planes[1], cam->planes[2], cam->planes[3]};
    F.time = glob->time;
    F.use_shadowmaps = glob->use_shadowmaps;
    F.shadow = o.shadow.data();
    o.cnt = Counters();
    std::vector<Counters> tcnt((size_t)std::max(1, threads));

    Model& G = sc.models[WR_MODEL_GEOMETRY];
    const M4 gm = to_m4(G.inst.model);
    M3 gn; memcpy(gn.c, G.inst.normal, 36);
    size_t nv = G.verts.size(), nt = G.idx.size() / 3;

    // ---- shadow passes, state.rs:633-675 with depth.wgsl:3-17
    o.n_subtris_shadow = 0;
    if (passes & WR_PASS_SHADOWS) {
        std::vector<V4> wp(nv);
        for (size_t i = 0; i < nv; ++i) {
            const float* p = G.verts[i].position;
            wp[i] = mul(gm, V4{p[0], p[1], p[2], 1.0f});
        }
        Target tg{o.S, o.S};
        for (int face = 0; face < 6; ++face) {
            float* L = o.shadow.data() + (size_t)face * o.S * o.S;
            std::fill(L, L + (size_t)o.S * o.S, 1.0f);
            std::vector<V4> clip(nv);
            for (size_t i = 0; i < nv; ++i) clip[i] = mul(F.light_m[face], wp[i]);
            std::vector<SubTri> subs;
            for (size_t t = 0; t < nt; ++t) {
                V4 c[3] = {clip[G.idx[3 * t]], clip[G.idx[3 * t + 1]], clip[G.idx[3 * t + 2]]};
                setup_triangle(c, (uint32_t)t, tg, true, subs);
            }
            o.n_subtris_shadow += subs.size();
            parallel_bands(o.S, threads, [&](int, int y0, int y1) {
                for (const SubTri& st : subs) {
                    int ya = std::max(st.pymin, y0), yb = std::min(st.pymax, y1 - 1);
                    for (int py = ya; py <= yb; ++py)
                        for (int px = st.pxmin; px <= st.pxmax; ++px) {
                            int64_t E[3];
                            if (!covers(st, px, py, E)) continue;
                            float z = frag_depth(st, E);
                            float zb = fminf(fmaxf(z + st.bias, 0.0f), 1.0f);
                            float& d = L[(size_t)py * o.S + px];
                            if (zb <= d) d = zb;  // pass.rs:118 LessEqual
                        }
                }
            });
        }
    }

    // ---- geometry pass, state.rs:688-726 with pbr.wgsl
    std::fill(o.depth.begin(), o.depth.end(), 1.0f);
    std::fill(o.vis.begin(), o.vis.end(), 0xFFFFFFFFu);
    for (size_t i = 0; i < (size_t)o.W * o.H; ++i) { o.color[4 * i] = 0; o.color[4 * i + 1] = 0; o.color[4 * i + 2] = 0; o.color[4 * i + 3] = 255; }
    Target tg{o.W, o.H};
    std::vector<Varyings> vs(nv);
    for (size_t i = 0; i < nv; ++i) {  // pbr.wgsl:8-46
        const WrModelVertex& v = G.verts[i];
        V3 wn = normalize3(mul(gn, v3(v.normal[0], v.normal[1], v.normal[2])));
        V3 wt = normalize3(mul(gn, v3(v.tangent[0], v.tangent[1], v.tangent[2])));
        V3 wb = normalize3(mul(gn, v3(v.bitangent[0], v.bitangent[1], v.bitangent[2])));
        V4 wpos = mul(gm, V4{v.position[0], v.position[1], v.position[2], 1.0f});
        Varyings& out = vs[i];
        out.clip = mul(F.pv, wpos);
        out.uv = V2{v.tex_coords[0], v.tex_coords[1]};
        V3 w3 = v3(wpos.x, wpos.y, wpos.z);
        out.tpos = v3(dot3(wt, w3), dot3(wb, w3), dot3(wn, w3));
        out.tlight = v3(dot3(wt, F.light_pos), dot3(wb, F.light_pos), dot3(wn, F.light_pos));
        out.tview = v3(dot3(wt, F.cam_pos), dot3(wb, F.cam_pos), dot3(wn, F.cam_pos));
        out.wpos = wpos;
    }
    std::vector<SubTri> subs;
    for (size_t t = 0; t < nt; ++t) {
        V4 c[3] = {vs[G.idx[3 * t]].clip, vs[G.idx[3 * t + 1]].clip, vs[G.idx[3 * t + 2]].clip};
        setup_triangle(c, (uint32_t)t, tg, false, subs);
    }
    o.n_subtris_geom = subs.size();
    parallel_bands(o.H, threads, [&](int tid, int y0, int y1) {
        Counters& cn = tcnt[tid];
        for (const SubTri& st : subs) {
            int ya = std::max(st.pymin, y0), yb = std::min(st.pymax, y1 - 1);
            if (ya > yb) continue;
            const Varyings& a0 = vs[G.idx[3 * st.tri]];
            const Varyings& a1 = vs[G.idx[3 * st.tri + 1]];
            const Varyings& a2 = vs[G.idx[3 * st.tri + 2]];
            const Material& mat = G.mats[G.tri_mat[st.tri]];
            for (int py = ya; py <= yb; ++py)
                for (int px = st.pxmin; px <= st.pxmax; ++px) {
                    int64_t E[3];
                    if (!covers(st, px, py, E)) continue;
                    float z = frag_depth(st, E);
                    size_t pi = (size_t)py * o.W + px;
                    if (!(z < o.depth[pi])) continue;  // pass.rs:118 Less
                    o.depth[pi] = z;
                    o.vis[pi] = (st.tri << 3) | st.sub;
                    float B[3];
                    frag_bary(st, E, B);
                    V2 uv{ip(B, a0.uv.x, a1.uv.x, a2.uv.x), ip(B, a0.uv.y, a1.uv.y, a2.uv.y)};
                    // R12 quad derivatives
                    int qx = px & ~1, qy = py & ~1;
#ifdef WRO_ALT_R12_FINE_DERIVATIVES
                    // R12 alternative: ddx along the fragment's own row, ddy along its own column
                    V2 q[4];
                    const int qp[4][2] = {{qx, py}, {qx + 1, py}, {px, qy}, {px, qy + 1}};
                    for (int k = 0; k < 4; ++k) {
                        int64_t Eq[3]; float Bq[3];
                        edge_values(st, qp[k][0], qp[k][1], Eq);
                        frag_bary(st, Eq, Bq);
                        q[k] = V2{ip(Bq, a0.uv.x, a1.uv.x, a2.uv.x), ip(Bq, a0.uv.y, a1.uv.y, a2.uv.y)};
                    }
                    V2 ddx{q[1].x - q[0].x, q[1].y - q[0].y}, ddy{q[3].x - q[2].x, q[3].y - q[2].y};
#else
                    V2 q[3];
                    const int qo[3][2] = {{0, 0}, {1, 0}, {0, 1}};
                    for (int k = 0; k < 3; ++k) {
                        int64_t Eq[3]; float Bq[3];
                        edge_values(st, qx + qo[k][0], qy + qo[k][1], Eq);
                        frag_bary(st, Eq, Bq);
                        q[k] = V2{ip(Bq, a0.uv.x, a1.uv.x, a2.uv.x), ip(Bq, a0.uv.y, a1.uv.y, a2.uv.y)};
                    }
                    V2 ddx{q[1].x - q[0].x, q[1].y - q[0].y}, ddy{q[2].x - q[0].x, q[2].y - q[0].y};
#endif
                    V3 tpos = v3(ip(B, a0.tpos.x, a1.tpos.x, a2.tpos.x), ip(B, a0.tpos.y, a1.tpos.y, a2.tpos.y), ip(B, a0.tpos.z, a1.tpos.z, a2.tpos.z));
                    V3 tlight = v3(ip(B, a0.tlight.x, a1.tlight.x, a2.tlight.x), ip(B, a0.tlight.y, a1.tlight.y, a2.tlight.y), ip(B, a0.tlight.z, a1.tlight.z, a2.tlight.z));
                    V3 tview = v3(ip(B, a0.tview.x, a1.tview.x, a2.tview.x), ip(B, a0.tview.y, a1.tview.y, a2.tview.y), ip(B, a0.tview.z, a1.tview.z, a2.tview.z));
                    V4 wpos{ip(B, a0.wpos.x, a1.wpos.x, a2.wpos.x), ip(B, a0.wpos.y, a1.wpos.y, a2.wpos.y), ip(B, a0.wpos.z, a1.wpos.z, a2.wpos.z), ip(B, a0.wpos.w, a1.wpos.w, a2.wpos.w)};
                    V4 src = fs_pbr(sc, mat, uv, ddx, ddy, tpos, tlight, tview, wpos);
                    cn.shaded_fragments++;
                    if (!(src.w == 1.0f)) cn.blended_lt1++;
                    blend_store(&o.color[4 * pi], src, true);  // state.rs:466 has_transparency
                }
        }
    });

    // ---- light debug pass, state.rs:728-759 with light_debug.wgsl:12-26
    if (passes & WR_PASS_LIGHT_DEBUG) {
        Model& Lm = sc.models[WR_MODEL_LIGHT_MARKER];
        size_t lnv = Lm.verts.size(), lnt = Lm.idx.size() / 3;
        std::vector<V4> clip(lnv);
        for (size_t i = 0; i < lnv; ++i) {
            const float* p = Lm.verts[i].position;
            V4 wp{p[0] * 10.0f + F.light_pos.x, p[1] * 10.0f + F.light_pos.y, p[2] * 10.0f + F.light_pos.z, 1.0f};
            clip[i] = mul(F.pv, wp);
        }
        std::vector<SubTri> ls;
        for (size_t t = 0; t < lnt; ++t) {
            V4 c[3] = {clip[Lm.idx[3 * t]], clip[Lm.idx[3 * t + 1]], clip[Lm.idx[3 * t + 2]]};
            setup_triangle(c, (uint32_t)(nt + t), tg, false, ls);
        }
        V4 src{F.light_color.x, F.light_color.y, F.light_color.z, 1.0f};
        for (const SubTri& st : ls)
            for (int py = st.pymin; py <= st.pymax; ++py)
                for (int px = st.pxmin; px <= st.pxmax; ++px) {
                    int64_t E[3];
                    if (!covers(st, px, py, E)) continue;
                    float z = frag_depth(st, E);
                    size_t pi = (size_t)py * o.W + px;
                    if (!(z < o.depth[pi])) continue;
                    o.depth[pi] = z;  // state.rs:482 write_depth = true
                    o.vis[pi] = (st.tri << 3) | st.sub;
                    blend_store(&o.color[4 * pi], src, false);  // state.rs:481 opaque
                }
    }

    // ---- fog pass, state.rs:763-804 with fog.wgsl
    if (passes & WR_PASS_FOG) {
        Model& Fm = sc.models[WR_MODEL_FOG_VOLUME];
        const M4 fm = to_m4(Fm.inst.model);
        size_t fnv = Fm.verts.size(), fnt = Fm.idx.size() / 3;
        std::vector<V4> wp(fnv), clip(fnv);
        for (size_t i = 0; i < fnv; ++i) {  // fog.wgsl:14-34
            const float* p = Fm.verts[i].position;
            wp[i] = mul(fm, V4{p[0], p[1], p[2], 1.0f});
            clip[i] = mul(F.pv, wp[i]);
        }
        std::vector<SubTri> fs;
        for (size_t t = 0; t < fnt; ++t) {
            V4 c[3] = {clip[Fm.idx[3 * t]], clip[Fm.idx[3 * t + 1]], clip[Fm.idx[3 * t + 2]]};
            setup_triangle(c, (uint32_t)t, tg, false, fs);
        }
        const float* dtex = o.depth.data();
        parallel_bands(o.H, threads, [&](int tid, int y0, int y1) {
            Counters& cn = tcnt[tid];
            for (const SubTri& st : fs) {
                int ya = std::max(st.pymin, y0), yb = std::min(st.pymax, y1 - 1);
                const V4& w0 = wp[Fm.idx[3 * st.tri]];
                const V4& w1 = wp[Fm.idx[3 * st.tri + 1]];
                const V4& w2 = wp[Fm.idx[3 * st.tri + 2]];
                for (int py = ya; py <= yb; ++py)
                    for (int px = st.pxmin; px <= st.pxmax; ++px) {
                        int64_t E[3];
                        if (!covers(st, px, py, E)) continue;
                        float z = frag_depth(st, E);
                        size_t pi = (size_t)py * o.W + px;
                        if (!(z < o.depth[pi])) continue;  // Less, no depth write (state.rs:502)
                        float B[3];
                        frag_bary(st, E, B);
                        V3 origin = v3(ip(B, w0.x, w1.x, w2.x), ip(B, w0.y, w1.y, w2.y), ip(B, w0.z, w1.z, w2.z));
                        cn.fog_fragments++;
                        V4 src = fs_fog(sc, dtex, origin, z, px, py, cn);
                        blend_store(&o.color[4 * pi], src, true);  // state.rs:501 has_transparency
                    }
            }
        });
    }
    for (const Counters& c : tcnt) {
        o.cnt.fog_fragments += c.fog_fragments;
        o.cnt.fog_noise_evals += c.fog_noise_evals;
        o.cnt.fog_noise_evals_unused += c.fog_noise_evals_unused;
        o.cnt.shaded_fragments += c.shaded_fragments;
        o.cnt.blended_lt1 += c.blended_lt1;
    }
}

}  // namespace

// ============================================================================= C API
extern "C" {

void* wro_create(uint32_t w, uint32_t h, uint32_t shadow_size) {
    Oracle* o = new Oracle();
    o->W = (int)w; o->H = (int)h; o->S = (int)(shadow_size ? shadow_size : 2048);
    o->color.assign((size_t)w * h * 4, 0);
    o->depth.assign((size_t)w * h, 1.0f);
    o->vis.assign((size_t)w * h, 0xFFFFFFFFu);
    o->shadow.assign((size_t)6 * o->S * o->S, 1.0f);
    return o;
}
void wro_destroy(void* p) { delete (Oracle*)p; }

uint32_t wro_add_texture(void* p, const uint8_t* rgba, uint32_t w, uint32_t h, int srgb) {
    Oracle* o = (Oracle*)p;
    Texture t; t.w = w; t.h = h; t.srgb = srgb; t.px.assign(rgba, rgba + (size_t)w * h * 4);
    o->sc.tex.push_back(std::move(t));
    return (uint32_t)o->sc.tex.size() - 1;
}
uint32_t wro_add_material(void* p, uint32_t model, uint32_t tb, uint32_t tn, uint32_t tmr, float metallic, float roughness) {
    Oracle* o = (Oracle*)p;
    o->sc.models[model].mats.push_back(Material{tb, tn, tmr, metallic, roughness});
    return (uint32_t)o->sc.models[model].mats.size() - 1;
}
void wro_add_mesh(void* p, uint32_t model, const WrModelVertex* v, uint32_t nv, const uint32_t* idx, uint32_t ni, uint32_t material) {
    Oracle* o = (Oracle*)p;
    Model& m = o->sc.models[model];
    uint32_t base = (uint32_t)m.verts.size();
    m.verts.insert(m.verts.end(), v, v + nv);
    for (uint32_t i = 0; i + 2 < ni; i += 3) {
        m.idx.push_back(base + idx[i]); m.idx.push_back(base + idx[i + 1]); m.idx.push_back(base + idx[i + 2]);
        m.tri_mat.push_back(material);
    }
}
void wro_set_instance(void* p, uint32_t model, const WrInstanceRaw* inst) {
    Oracle* o = (Oracle*)p;
    o->sc.models[model].inst = *inst;
    o->sc.models[model].has_inst = true;
}
int wro_render(void* p, const WrCameraUniform* cam, const WrLightUniform* light, const WrGlobalUniforms* glob, uint32_t passes, int threads) {
    Oracle* o = (Oracle*)p;
    if (!o->sc.models[0].has_inst) return -6;
    if ((passes & WR_PASS_FOG) && !o->sc.models[1].has_inst) return -6;
    render(*o, cam, light, glob, passes, threads);
    return 0;
}
const uint8_t* wro_color(void* p) { return ((Oracle*)p)->color.data(); }
const float* wro_depth(void* p) { return ((Oracle*)p)->depth.data(); }
const uint32_t* wro_visibility(void* p) { return ((Oracle*)p)->vis.data(); }
const float* wro_shadow(void* p) { return ((Oracle*)p)->shadow.data(); }
void wro_counters(void* p, uint64_t out[8]) {
    Oracle* o = (Oracle*)p;
    out[0] = o->cnt.fog_fragments; out[1] = o->cnt.fog_noise_evals; out[2] = o->cnt.shaded_fragments;
    out[3] = o->cnt.blended_lt1; out[4] = o->n_subtris_geom; out[5] = o->n_subtris_shadow; out[6] = o->cnt.fog_noise_evals_unused; out[7] = 0;
}

// ---- unit functions for known-answer tests
float wro_sin(float x) { return o_sin(x); }
void wro_sin_many(const float* x, int n, float* out) { for (int i = 0; i < n; ++i) out[i] = o_sin(x[i]); }
// rand of noise.wgsl:3-5 from the value of its dot product
void wro_hash_many(const float* d, int n, float* out) { for (int i = 0; i < n; ++i) out[i] = o_hash_of_dot(d[i]); }
// Floats d with bit patterns in [lo_bits, hi_bits) (positive d; both signs are tried) for which
// the hash's s - floor(s) is not below 1, i.e. where a hardware fraction instruction (which
// returns the float below 1 there) would differ from the written formula.
uint64_t wro_hash_fract_exceptions(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, uint32_t phase) {
    uint64_t bad = 0;
    for (uint64_t b = (uint64_t)lo_bits + phase; b < hi_bits; b += stride)
        for (uint32_t sg = 0; sg < 2; ++sg) {
            float s = o_sin(u2f((uint32_t)b | (sg << 31))) * 42123.45f;
            if (!(s - floorf(s) < 1.0f)) ++bad;
        }
    return bad;
}
float wro_rand(const float p[3]) { return o_rand(v3(p[0], p[1], p[2])); }
float wro_value_noise(const float p[3]) { return o_value_noise(v3(p[0], p[1], p[2])); }
float wro_fbm(const float p[3]) { return o_fbm(v3(p[0], p[1], p[2])); }
float wro_fog_noise(const float p[3], float time) { return o_fog_noise(v3(p[0], p[1], p[2]), time); }
float wro_depth_to_linear(float d, float zn, float zf) { return depth_to_linear(d, zn, zf); }
uint8_t wro_srgb_encode(float x) { return srgb_encode(x); }
float wro_srgb_decode(uint8_t c) { return g_srgb.decode[c]; }
void wro_brdf(const float n[3], const float l[3], const float v[3], const float h[3], const float albedo[3], float roughness, float metalness, float out[3]) {
    V3 r = brdf(v3(n[0], n[1], n[2]), v3(l[0], l[1], l[2]), v3(v[0], v[1], v[2]), v3(h[0], h[1], h[2]), v3(albedo[0], albedo[1], albedo[2]), roughness, metalness);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void wro_ambient(const float light[4], float dist, float sld, float out[3]) {
    V3 r = sample_ambient_light(V4{light[0], light[1], light[2], light[3]}, dist, sld);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// Sets up one triangle given in clip space.  Each sub-triangle writes 24 floats/ints:
// X0,Y0,X1,Y1,X2,Y2 (as float-converted ints are NOT used; ints are bit-copied), z0,dz1,dz2,
// inv2A, invw0..2, bias, sub, pxmin,pymin,pxmax,pymax, then 9 clip weights -> 28 words.
int wro_setup_triangle(const float clip[12], uint32_t W, uint32_t H, int shadow, uint32_t* out, int max_subs) {
    V4 c[3];
    for (int k = 0; k < 3; ++k) c[k] = V4{clip[4 * k], clip[4 * k + 1], clip[4 * k + 2], clip[4 * k + 3]};
    std::vector<SubTri> subs;
    setup_triangle(c, 0, Target{(int)W, (int)H}, shadow != 0, subs);
    int n = 0;
    for (const SubTri& st : subs) {
        if (n >= max_subs) break;
        uint32_t* w = out + (size_t)n * 28;
        for (int k = 0; k < 3; ++k) { w[2 * k] = (uint32_t)st.X[k]; w[2 * k + 1] = (uint32_t)st.Y[k]; }
        w[6] = f2u(st.z0); w[7] = f2u(st.dz1); w[8] = f2u(st.dz2); w[9] = f2u(st.inv2A);
        w[10] = f2u(st.invw[0]); w[11] = f2u(st.invw[1]); w[12] = f2u(st.invw[2]); w[13] = f2u(st.bias);
        w[14] = st.sub; w[15] = (uint32_t)st.pxmin; w[16] = (uint32_t)st.pymin; w[17] = (uint32_t)st.pxmax; w[18] = (uint32_t)st.pymax;
        for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) w[19 + 3 * k + j] = f2u(st.beta[k][j]);
        ++n;
    }
    return n;
}

}  // extern "C"
