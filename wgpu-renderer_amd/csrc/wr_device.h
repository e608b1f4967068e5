// wr_device.h -- per-vertex / per-triangle / per-fragment arithmetic of the HIP render path.
//
// Everything here is `__host__ __device__` so the same source is what the gfx950 kernels
// execute and what a host-side unit check can call; the kernels that schedule this work
// live in wr_kernels.h.  The arithmetic restates the reference's WGSL programs
// (res/shaders/*.wgsl) and the fixed-function behaviour its pipelines request
// (src/core/pass.rs:52-135); citations sit on each function.
//
// Numerical contract (the library is built with -ffp-contract=off):
//   * f32 throughout, one rounding per written operation, fused multiply-add only where
//     fmaf is spelled out (the software sine);
//   * mat*vec = ((c0*v0 + c1*v1) + c2*v2) + c3*v3, dot = (x*x' + y*y') + z*z';
//   * normalize(v) = v * (1/sqrt(dot(v,v))); mix(a,b,t) = a*(1-t) + b*t;
//   * coverage is decided on exact 64-bit integer edge functions over vertices snapped to
//     1/256 pixel, with the top-left rule; depth and barycentrics derive from those
//     integers, so raster, shading and the ordered-blend replay all see identical values.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define WR_HD __host__ __device__ inline

namespace wr {

constexpr int kBin = 64;             // screen bin edge in pixels (one workgroup, one LDS tile)
constexpr float kGuard = 8.0f;       // guard band: |x|,|y| <= kGuard * w survives unclipped
constexpr uint32_t kRecInvalid = 0xFFFFFFFFu;
constexpr uint32_t kRecClippedParent = 0xFFFFFFFEu;
constexpr uint32_t kVisNone = 0xFFFFFFFFu;
constexpr int kMaxFan = 7;           // 3 + 6 clip planes -> 9 vertices -> 7 fan triangles

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

WR_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
WR_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
WR_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
WR_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
WR_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
WR_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
WR_HD V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
WR_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
WR_HD float length3(V3 a) { return sqrtf(dot3(a, a)); }
WR_HD V3 normalize3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }
WR_HD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
WR_HD float lerpf(float a, float b, float f) { return a + f * (b - a); }
WR_HD float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
WR_HD int imin(int a, int b) { return a < b ? a : b; }
WR_HD int imax(int a, int b) { return a > b ? a : b; }
WR_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
WR_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// column-major 4x4, m[4*col + row]
WR_HD V4 mat4_mul(const float* m, V4 v) {
    V4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
WR_HD V3 mat3_mul(const float* m, V3 v) {  // column-major 3x3, m[3*col + row]
    V3 r;
    r.x = (m[0] * v.x + m[3] * v.y) + m[6] * v.z;
    r.y = (m[1] * v.x + m[4] * v.y) + m[7] * v.z;
    r.z = (m[2] * v.x + m[5] * v.y) + m[8] * v.z;
    return r;
}
inline void mat4_mul_mat4(const float* a, const float* b, float* out) {  // out = a*b, host side
    for (int j = 0; j < 4; ++j) {
        V4 c = mat4_mul(a, V4{b[4 * j], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3]});
        out[4 * j] = c.x; out[4 * j + 1] = c.y; out[4 * j + 2] = c.z; out[4 * j + 3] = c.w;
    }
}

// Software sine of the fog noise hash (noise.wgsl:4).  WGSL leaves the accuracy of sin to the
// implementation; the driver the reference is judged on (Mesa llvmpipe/lavapipe) lowers it to
// the single-precision Cephes scheme (as in sse_mathfun's sin_ps), so that is the contract:
//   j   = octant of |x|: trunc(|x| * 4/pi), rounded up to even,
//   r   = |x| - j*pi/4, pi/4 taken in three parts (0.78515625 + 2.4187564849853515625e-4 +
//         3.77489497744594108e-8) by fused multiply-adds,
//   j&2 = 0: sin(r) = r + r*z*(s2 + z*(s1 + z*s0)), z = r*r
//   j&2 = 2: cos(r) = 1 - z/2 + z*z*(c2 + z*(c1 + z*c0))
//   sign = sign(x) xor (j&4).
// |x| >= 2^30 (where trunc no longer fits the octant arithmetic) or NaN -> +0.  GUARD = false
// is for callers that have already bounded |x| below 2^30; for such x both forms return the
// same bits.  On the device both polynomials are evaluated and one is selected (two dependent
// chains that fill each other's latency; a branch would diverge on every hash).
constexpr float kSinRange = 1073741824.0f;
// GUARD selects one of three forms that return the same bits wherever they are allowed:
//   kSinGuarded  any argument;
//   kSinInt      |x| < 2^30 promised by the caller: no range test;
//   kSinFast     |x| <= kSinFastRange promised by the caller: the octant arithmetic stays in
//                floating point.  h = j/2 = floor(|x| * 4/pi * 0.5 + 0.5) (the sum is exact
//                while the product is below 2^24), r = |x| - h * (pi/2 in three parts) (the
//                same products and roundings, the constants doubled), polynomial from the
//                parity of h, sign from its second bit.  Three instructions fewer per hash;
//                compared with kSinGuarded on every float of the range by the CPU tests.
constexpr int kSinInt = 0, kSinGuarded = 1, kSinFast = 2;
constexpr float kSinFastRange = 13176794.0f;  // 2^24 / (4/pi), rounded down
template <int GUARD>
WR_HD float wr_sin_t(float x) {
    float ax = fabsf(x);
    uint32_t sign = f2u(x) & 0x80000000u;
    if (GUARD == kSinGuarded) {
        // Branch-free: an out-of-range (or NaN) argument is replaced by +0 on entry, for
        // which the arithmetic below yields +0.
        const bool in_range = ax < kSinRange;
        ax = in_range ? ax : 0.0f;
        sign = in_range ? sign : 0u;
    }
    const float y0 = ax * 1.27323954473516f;
    float r;
    uint32_t half_octant;  // j / 2: bit 0 selects the polynomial, bit 1 flips the sign
    if (GUARD == kSinFast) {
        const float h = floorf(fmaf(y0, 0.5f, 0.5f));
        half_octant = (uint32_t)(int)h;
        r = fmaf(h, -1.5703125f, ax);
        r = fmaf(h, -4.837512969970703125e-4f, r);
        r = fmaf(h, -7.54978995489188216e-8f, r);
    } else {
        int j = (int)y0;
        j = (j + 1) & ~1;
        const float y = (float)j;
        half_octant = (uint32_t)j >> 1;
        r = fmaf(y, -0.78515625f, ax);
        r = fmaf(y, -2.4187564849853515625e-4f, r);
        r = fmaf(y, -3.77489497744594108e-8f, r);
    }
    const float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = ps * z;
    ps = fmaf(ps, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = pc * z;
    pc = pc * z;
    pc = fmaf(z, -0.5f, pc);
    pc = pc + 1.0f;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WR_SIN_SELECT_BY_COMPARE)
    // bit 0 of the half octant, spread over a word, picks the polynomial bit by bit: a bit-field
    // extract and a bit select where a test, a compare and a conditional move were (one
    // instruction less in each of the eight hashes of an octave)
    uint32_t pick;  // (as assembly: the optimiser turns the plain expression back into the test and compare)
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(pick) : "v"(half_octant));
    const uint32_t res = (f2u(pc) & pick) | (f2u(ps) & ~pick);
    return u2f(res ^ ((sign ^ (half_octant << 30)) & 0x80000000u));
#else
    const float res = (half_octant & 1u) ? pc : ps;
    return u2f(f2u(res) ^ ((sign ^ (half_octant << 30)) & 0x80000000u));
#endif
}
WR_HD float wr_sin(float x) { return wr_sin_t<true>(x); }

// ---------------------------------------------------------------------------------------
// Per-frame constants, passed to kernels by value.
struct FrameConsts {
    float pv[16];            // camera.proj * camera.view (pbr.wgsl:37, one product per frame)
    float model_geom[16];    // InstanceRaw.model of the geometry model
    float normal_geom[9];    // InstanceRaw.normal
    float model_fog[16];
    float light_m[6][16];    // LightUniform.matrices
    float cam_pos[3];
    float light_pos[3];
    float light_color[4];
    float planes[4];         // near, far, width, height (camera.rs:133)
    float time;
    float drift_xy;          // time * 0.2            (fog.wgsl:50-51)
    float drift_z;           // sin(time*0.1) * 0.5   (fog.wgsl:52)
    uint32_t use_shadowmaps;
    uint32_t light_tame;     // every entry of light_m is finite and below 1e15 in magnitude
    int W, H, S;
    int tile_rank, tile_world;
    // Which bins the geometry pass (visibility + depth) covers: the context's own, or all of
    // them (0, 1) when the fog pass's depth lookups reach into neighbouring bins.
    int geom_tile_rank, geom_tile_world;
};

// ---------------------------------------------------------------------------------------
// Set-up triangle record: 64 bytes, read wave-uniformly by the rasteriser.
struct alignas(16) Rec {
    int32_t X0, Y0, X1, Y1;
    int32_t X2, Y2;
    float z0, dz1;
    float dz2, inv2A, iw0, iw1;   // shadow records keep the depth bias in iw0
    float iw2;
    uint32_t id;                  // (triangle id << 3) | fan index, or a kRec* marker
    uint32_t bbmin, bbmax;        // pixel bounding box, x | y << 16
};
static_assert(sizeof(Rec) == 64, "Rec must be 64 bytes");

struct ClipVert { float x, y, z, w, b0, b1, b2; };

WR_HD float plane_dist(const ClipVert& v, int plane) {
    switch (plane) {
        case 0: return v.z;
        case 1: return v.w - v.z;
        case 2: return v.x + kGuard * v.w;
        case 3: return kGuard * v.w - v.x;
        case 4: return v.y + kGuard * v.w;
        default: return kGuard * v.w - v.y;
    }
}
WR_HD ClipVert clip_lerp(const ClipVert& in, const ClipVert& out, float t) {
    ClipVert r;
    r.x = in.x + t * (out.x - in.x);
    r.y = in.y + t * (out.y - in.y);
    r.z = in.z + t * (out.z - in.z);
    r.w = in.w + t * (out.w - in.w);
    r.b0 = in.b0 + t * (out.b0 - in.b0);
    r.b1 = in.b1 + t * (out.b1 - in.b1);
    r.b2 = in.b2 + t * (out.b2 - in.b2);
    return r;
}

// Sutherland-Hodgman against z>=0, z<=w and the four guard-band planes, in that order.
// New vertices are always interpolated from the inside end of the edge so that an edge
// shared by two triangles clips to the same point from either side.
// wgpu clips to the Vulkan volume 0<=z<=w (pass.rs:111 unclipped_depth: false).
// (`tmp`: room for ten vertices; a kernel that clips a handful of triangles keeps both arrays in
// LDS, where the many dependent accesses of the loops below cost a tenth of what they cost in
// per-lane scratch memory.)
__host__ __device__ __attribute__((noinline)) inline int clip_polygon_in(ClipVert* poly, ClipVert* tmp, int n) {
    for (int plane = 0; plane < 6; ++plane) {
        bool any_out = false;
        for (int i = 0; i < n; ++i) if (!(plane_dist(poly[i], plane) >= 0.0f)) any_out = true;
        if (!any_out) continue;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            ClipVert a = poly[i];
            ClipVert b = poly[(i + 1 == n) ? 0 : i + 1];
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
__host__ __device__ inline int clip_polygon(ClipVert* poly, int n) {
    ClipVert tmp[10];
    return clip_polygon_in(poly, tmp, n);
}

// Viewport transform, sub-pixel snap, cull, bounding box, depth plane and (shadow) bias of
// one triangle given by three clip-space vertices.  beta receives the clip weights of the
// three set-up vertices in their final order.  Returns false if nothing can be covered.
// Fixed-function state: pass.rs:103-127 (CCW front, cull Back, depth bias {2, 2.0, 0}).
WR_HD bool setup_sub(const ClipVert* cv, int W, int H, bool shadow, uint32_t id, Rec& rec, float* beta) {
    float sx = 0.5f * (float)W, tx = 0.5f * (float)W;
    float sy = -0.5f * (float)H, ty = 0.5f * (float)H;
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
    if (area2 >= 0) return false;  // back-facing (y-down window) or degenerate
    // vertex order (0,2,1) makes the area positive
    rec.X0 = X[0]; rec.Y0 = Y[0]; rec.X1 = X[2]; rec.Y1 = Y[2]; rec.X2 = X[1]; rec.Y2 = Y[1];
    float z0 = zw[0], z1 = zw[2], z2 = zw[1];
    rec.iw0 = iw[0]; rec.iw1 = iw[2]; rec.iw2 = iw[1];
    if (beta) {
        beta[0] = cv[0].b0; beta[1] = cv[0].b1; beta[2] = cv[0].b2;
        beta[3] = cv[2].b0; beta[4] = cv[2].b1; beta[5] = cv[2].b2;
        beta[6] = cv[1].b0; beta[7] = cv[1].b1; beta[8] = cv[1].b2;
    }
    int xmin = imin(rec.X0, imin(rec.X1, rec.X2)), xmax = imax(rec.X0, imax(rec.X1, rec.X2));
    int ymin = imin(rec.Y0, imin(rec.Y1, rec.Y2)), ymax = imax(rec.Y0, imax(rec.Y1, rec.Y2));
    int pxmin = imax((xmin - 128 + 255) >> 8, 0), pymin = imax((ymin - 128 + 255) >> 8, 0);
    int pxmax = imin((xmax - 128) >> 8, W - 1), pymax = imin((ymax - 128) >> 8, H - 1);
    if (pxmin > pxmax || pymin > pymax) return false;
    rec.bbmin = (uint32_t)pxmin | ((uint32_t)pymin << 16);
    rec.bbmax = (uint32_t)pxmax | ((uint32_t)pymax << 16);
    rec.inv2A = 1.0f / (float)(-area2);
    rec.z0 = z0;
    rec.dz1 = z1 - z0;
    rec.dz2 = z2 - z0;
    rec.id = id;
    if (shadow) {
        // Vulkan depth-bias rule for a float depth format: slope term from the depth plane,
        // constant term in units of 2^(exponent(max z) - 23).
        float A1 = (float)(rec.Y2 - rec.Y0), A2 = (float)(rec.Y0 - rec.Y1);
        float B1 = (float)(rec.X0 - rec.X2), B2 = (float)(rec.X1 - rec.X0);
        float k = 256.0f * rec.inv2A;
        float dzdx = (A1 * rec.dz1 + A2 * rec.dz2) * k;
        float dzdy = (B1 * rec.dz1 + B2 * rec.dz2) * k;
        float m = fmaxf(fabsf(dzdx), fabsf(dzdy));
        float zmax = fmaxf(fmaxf(z0, z1), fmaxf(z2, 0.0f));
        uint32_t e8 = (f2u(zmax) >> 23) & 0xffu;
        float r = e8 >= 24u ? u2f((e8 - 23u) << 23) : 0.0f;
        rec.iw0 = m * 2.0f + r * 2.0f;
    }
    return true;
}

// Frustum classification of a triangle in clip space: 0 = drop, 1 = set up directly,
// 2 = needs the clipper.
WR_HD int classify_triangle(const V4* c) {
    bool r0 = true, r1 = true, r2 = true, r3 = true, r4 = true, r5 = true, need = false;
    for (int k = 0; k < 3; ++k) {
        V4 v = c[k];
        if (!(fabsf(v.x) < INFINITY) || !(fabsf(v.y) < INFINITY) || !(fabsf(v.z) < INFINITY) ||
            !(fabsf(v.w) < INFINITY)) return 0;  // any inf/nan coordinate
        if (!(v.x < -v.w)) r0 = false;
        if (!(v.x > v.w)) r1 = false;
        if (!(v.y < -v.w)) r2 = false;
        if (!(v.y > v.w)) r3 = false;
        if (!(v.z < 0.0f)) r4 = false;
        if (!(v.z > v.w)) r5 = false;
        float g = kGuard * v.w;
        if (v.z < 0.0f || v.z > v.w || v.x < -g || v.x > g || v.y < -g || v.y > g) need = true;
    }
    if (r0 || r1 || r2 || r3 || r4 || r5) return 0;
    return need ? 2 : 1;
}

// ---------------------------------------------------------------------------------------
// Edge functions of a record.  Edge i is opposite vertex i and positive inside.
struct Edges {
    int32_t A[3], B[3];
    int64_t C[3];
    int32_t tlb[3];  // 0 if the edge is top or left (points on it are inside), else -1
};
WR_HD void make_edges(const Rec& r, Edges& e) {
    int32_t X[3] = {r.X0, r.X1, r.X2}, Y[3] = {r.Y0, r.Y1, r.Y2};
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, b = (i + 2) % 3;
        e.A[i] = Y[a] - Y[b];
        e.B[i] = X[b] - X[a];
        e.C[i] = (int64_t)X[a] * (int64_t)Y[b] - (int64_t)X[b] * (int64_t)Y[a];
        e.tlb[i] = ((e.A[i] > 0) || (e.A[i] == 0 && e.B[i] > 0)) ? 0 : -1;
    }
}
WR_HD void edge_eval(const Edges& e, int px, int py, int64_t* E) {
    int32_t sx = px * 256 + 128, sy = py * 256 + 128;
    for (int i = 0; i < 3; ++i) E[i] = (int64_t)e.A[i] * sx + (int64_t)e.B[i] * sy + e.C[i];
}
WR_HD bool edges_inside(const Edges& e, const int64_t* E) {
    return ((E[0] + e.tlb[0]) | (E[1] + e.tlb[1]) | (E[2] + e.tlb[2])) >= 0;
}
// Largest value an edge takes over the pixel-centre samples of [x0,x1]x[y0,y1]; the box
// cannot touch the triangle if this (plus the fill-rule offset) is negative for any edge.
WR_HD bool box_rejected(const Edges& e, int x0, int y0, int x1, int y1) {
    int32_t sxa = x0 * 256 + 128, sxb = x1 * 256 + 128, sya = y0 * 256 + 128, syb = y1 * 256 + 128;
    for (int i = 0; i < 3; ++i) {
        int64_t m = (int64_t)e.A[i] * (e.A[i] > 0 ? sxb : sxa) + (int64_t)e.B[i] * (e.B[i] > 0 ? syb : sya) + e.C[i];
        if (m + e.tlb[i] < 0) return true;
    }
    return false;
}
WR_HD float depth_at(const Rec& r, const int64_t* E) {
    float l1 = (float)E[1] * r.inv2A, l2 = (float)E[2] * r.inv2A;
    float z = (r.z0 + l1 * r.dz1) + l2 * r.dz2;
    return fminf(fmaxf(z, 0.0f), 1.0f);
}
// Perspective-correct barycentrics with respect to the ORIGINAL triangle's vertices
// (index order of the index buffer).  beta == nullptr for an unclipped triangle, whose
// set-up order is (v0, v2, v1).
WR_HD void bary_at(const Rec& r, const float* beta, const int64_t* E, float* B) {
    float l1 = (float)E[1] * r.inv2A, l2 = (float)E[2] * r.inv2A;
    float l0 = (1.0f - l1) - l2;
    float p0 = l0 * r.iw0, p1 = l1 * r.iw1, p2 = l2 * r.iw2;
    float inv = 1.0f / ((p0 + p1) + p2);
    float q0 = p0 * inv, q1 = p1 * inv, q2 = p2 * inv;
    if (beta) {
        B[0] = (q0 * beta[0] + q1 * beta[3]) + q2 * beta[6];
        B[1] = (q0 * beta[1] + q1 * beta[4]) + q2 * beta[7];
        B[2] = (q0 * beta[2] + q1 * beta[5]) + q2 * beta[8];
    } else {
        B[0] = q0; B[1] = q2; B[2] = q1;
    }
}
WR_HD float ip(const float* B, float a0, float a1, float a2) { return (B[0] * a0 + B[1] * a1) + B[2] * a2; }

// ---------------------------------------------------------------------------------------
// Colour target Rgba8UnormSrgb.  tables = decode[256] | unorm[256] | thresh[256].
struct ColorTables {
    const float* t;
    WR_HD float decode(uint32_t c) const { return t[c]; }
    WR_HD float unorm(uint32_t c) const { return t[256 + c]; }
    WR_HD float thresh(uint32_t c) const { return t[512 + c]; }
};
// Largest code c with thresh(c) <= x.  On the device the search starts from an estimate
// (hardware log2/exp2 of the sRGB curve, a code or so off at worst) and walks to the code
// the thresholds define; the host build bisects.  Both return the same code.
WR_HD uint32_t srgb_encode(const ColorTables& ct, float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
#if defined(__HIP_DEVICE_COMPILE__)
    float est = x <= 0.0031308f ? x * 3294.6f : 269.025f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * 0.41666666f) - 14.025f;
    int g = (int)(est + 0.5f);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    while (g < 255 && x >= ct.thresh((uint32_t)(g + 1))) ++g;
    while (g > 0 && x < ct.thresh((uint32_t)g)) --g;
    return (uint32_t)g;
#else
    int lo = 0, hi = 255;
    for (int it = 0; it < 8; ++it) {
        int mid = (lo + hi + 1) >> 1;
        if (x >= ct.thresh((uint32_t)mid)) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
#endif
}
WR_HD uint32_t unorm_encode(float a) {
    if (!(a > 0.0f)) return 0;
    if (a >= 1.0f) return 255;
    return (uint32_t)(int)(a * 255.0f + 0.5f);
}
// pass.rs:67-83: SrcAlpha / OneMinusSrcAlpha on colour and alpha, on the decoded 8-bit
// destination; blend == false is BlendComponent::REPLACE.
WR_HD uint32_t blend_pack(const ColorTables& ct, uint32_t dst, V4 src, bool blend) {
    V4 o = src;
    if (blend) {
        float sa = src.w, ia = 1.0f - src.w;
        o.x = src.x * sa + ct.decode(dst & 255u) * ia;
        o.y = src.y * sa + ct.decode((dst >> 8) & 255u) * ia;
        o.z = src.z * sa + ct.decode((dst >> 16) & 255u) * ia;
        o.w = src.w * sa + ct.unorm(dst >> 24) * ia;
    }
    return srgb_encode(ct, o.x) | (srgb_encode(ct, o.y) << 8) | (srgb_encode(ct, o.z) << 16) | (unorm_encode(o.w) << 24);
}

// ---------------------------------------------------------------------------------------
// Material textures: texture.rs:118-126 (Repeat, mag Linear, min Nearest, one level).
struct TexDesc { uint32_t offset, w, h, srgb; };
struct MatDesc { uint32_t base, normal, mr; float metallic, roughness; uint32_t opaque /* base-colour alpha is 255 everywhere */, pad1, pad2; };

WR_HD int wrapi(int i, int n) { i %= n; return i < 0 ? i + n : i; }
WR_HD int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }
WR_HD float guard_coord(float x) { return (fabsf(x) < 1.0e9f) ? x : 0.0f; }

WR_HD V4 texel_fetch(const uint32_t* texels, const TexDesc& t, const ColorTables& ct, int x, int y) {
    uint32_t p = texels[t.offset + (uint32_t)y * t.w + (uint32_t)x];
    uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u, a = p >> 24;
    if (t.srgb) return V4{ct.decode(r), ct.decode(g), ct.decode(b), ct.unorm(a)};
    return V4{ct.unorm(r), ct.unorm(g), ct.unorm(b), ct.unorm(a)};
}
WR_HD V4 lerp4(V4 a, V4 b, float f) { return V4{lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f)}; }

// textureSample(t, s, uv) of pbr.wgsl:76-79.  The level of detail only chooses between the
// magnification (bilinear) and minification (nearest) filters; it comes from the per-quad
// derivatives of uv.
WR_HD V4 sample_material(const uint32_t* texels, const TexDesc& t, const ColorTables& ct, V2 uv, V2 ddx, V2 ddy) {
    float W = (float)t.w, H = (float)t.h;
    float rho = fmaxf(fmaxf(fabsf(ddx.x), fabsf(ddy.x)) * W, fmaxf(fabsf(ddx.y), fabsf(ddy.y)) * H);
    if (rho <= 1.0f) {
        float x = guard_coord(uv.x * W - 0.5f), y = guard_coord(uv.y * H - 0.5f);
        float x0 = floorf(x), y0 = floorf(y);
        float fx = x - x0, fy = y - y0;
        int i0 = wrapi((int)x0, (int)t.w), j0 = wrapi((int)y0, (int)t.h);
        int i1 = wrapi(i0 + 1, (int)t.w), j1 = wrapi(j0 + 1, (int)t.h);
        V4 top = lerp4(texel_fetch(texels, t, ct, i0, j0), texel_fetch(texels, t, ct, i1, j0), fx);
        V4 bot = lerp4(texel_fetch(texels, t, ct, i0, j1), texel_fetch(texels, t, ct, i1, j1), fx);
        return lerp4(top, bot, fy);
    }
    float x = guard_coord(floorf(uv.x * W)), y = guard_coord(floorf(uv.y * H));
    return texel_fetch(texels, t, ct, wrapi((int)x, (int)t.w), wrapi((int)y, (int)t.h));
}

// ---------------------------------------------------------------------------------------
// Shadow lookup: light.wgsl.
// The six faces' depth, then, in the same buffer, (min, max) depth of every 8x8 texel tile of
// every face (written with the faces by the shadow rasteriser): a lookup whose reference depth
// is not above the minimum, or is above the maximum, of the tiles its taps can touch knows
// every one of its compares without fetching a texel.
constexpr int kShadowTile = 8;
WR_HD int shadow_tiles_w(int S) { return (S + kShadowTile - 1) / kShadowTile; }
WR_HD size_t shadow_buffer_floats(int S) {
    return (size_t)6 * (size_t)S * (size_t)S + (size_t)12 * (size_t)shadow_tiles_w(S) * (size_t)shadow_tiles_w(S);
}
struct ShadowMap {
    const float* data;
    int S;
    WR_HD const float* tiles(int layer) const {
        return data + (size_t)6 * (size_t)S * (size_t)S + (size_t)2 * (size_t)layer * (size_t)shadow_tiles_w(S) * (size_t)shadow_tiles_w(S);
    }
};

// textureSampleCompare with the sampler of texture.rs:59-72 (ClampToEdge, Linear,
// compare LessEqual, state.rs:218): four compares, bilinear blend of the 0/1 results.
WR_HD float shadow_compare(const ShadowMap& sm, float u, float v, int layer, float ref) {
    float S = (float)sm.S;
    float x = guard_coord(u * S - 0.5f), y = guard_coord(v * S - 0.5f);
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int i0 = clampi((int)x0, sm.S), i1 = clampi((int)x0 + 1, sm.S);
    int j0 = clampi((int)y0, sm.S), j1 = clampi((int)y0 + 1, sm.S);
    const float* L = sm.data + (size_t)layer * (size_t)sm.S * (size_t)sm.S;
    float c00 = ref <= L[(size_t)j0 * sm.S + i0] ? 1.0f : 0.0f;
    float c10 = ref <= L[(size_t)j0 * sm.S + i1] ? 1.0f : 0.0f;
    float c01 = ref <= L[(size_t)j1 * sm.S + i0] ? 1.0f : 0.0f;
    float c11 = ref <= L[(size_t)j1 * sm.S + i1] ? 1.0f : 0.0f;
    return lerpf(lerpf(c00, c10, fx), lerpf(c01, c11, fx), fy);
}

// constants.wgsl:1-17, value for value (PI is the reference's truncated 3.14159 on purpose;
// INV_SHADOW_SAMPLES is 1/25 although the loops below take 16 taps).
constexpr float kPi = 3.14159f;
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt3 = 0.57735026918962576451f;
constexpr int kShadowSamples = 2;
constexpr float kInvShadowSamples = 1.0f / 25.0f;
constexpr int kFogMaxSteps = 30;
constexpr float kFogStepSize = 5.0f;
constexpr int kFogLightMaxSteps = 10;
constexpr float kFogLightStepSize = 10.0f;
constexpr float kFogDensity = 2.0f;
constexpr float kFogLightDensity = 3.0f;
constexpr float kFogAlpha = 1.0f;
constexpr float kFogDensityColorX = 0.5f, kFogDensityColorY = 0.1f;

// One axis of textureSampleCompare's footprint for the four taps of the loops below: the
// texel coordinate before clamping and the blend weight, worked out per tap exactly as
// shadow_compare does.
struct PcfAxis { int fl[4]; float fr[4]; };
WR_HD PcfAxis pcf_axis(float centre, float S, float ts) {
    PcfAxis a;
    for (int k = 0; k < 4; ++k) {
        float o = (float)(k - kShadowSamples) / ts;
        float x = guard_coord((centre + o) * S - 0.5f);
        float x0 = floorf(x);
        a.fr[k] = x - x0;
        a.fl[k] = (int)x0;
    }
    return a;
}
// light.wgsl:1-29.  The loops run x,y in [-2,2): 16 taps weighted 1/25 each.  The taps are one
// texel apart, so their sixteen 2x2 footprints are, all but never, the sixteen 2x2 windows of
// one 5x5 block of texels: that block is fetched and compared once (25 loads for 64), and each
// tap blends its four results with its own weights in the order of the loops.  Whether the
// footprints line up is checked on the tap coordinates themselves (rounding in (centre + o) * S
// can shift a tap's floor); if they do not, every tap does its own lookups.
// BLOCK = false always takes the tap-by-tap path: for kernels where the lookup is a small part
// of the work and the block's registers would cost a resident wavefront.
template <bool BLOCK>
WR_HD float sample_direct_light_index(const ShadowMap& sm, int index, V4 lc) {
    if (lc.w <= 0.0f) return 0.0f;
    float pc = 1.0f / lc.w;
    float lx = (lc.x * 0.5f) * pc + 0.5f;
    float ly = (lc.y * -0.5f) * pc + 0.5f;
    float ref = lc.z * pc - 0.000001f;
    float total = 0.0f;
    float ts = (float)sm.S;
    const PcfAxis ax = pcf_axis(lx, ts, ts), ay = pcf_axis(ly, ts, ts);
    // texels the sixteen footprints can touch, after clamping: columns [c0, c1], rows [r0, r1]
    int c0 = ax.fl[0], c1 = ax.fl[0], r0 = ay.fl[0], r1 = ay.fl[0];
    for (int k = 1; k < 4; ++k) {
        c0 = imin(c0, ax.fl[k]); c1 = imax(c1, ax.fl[k]);
        r0 = imin(r0, ay.fl[k]); r1 = imax(r1, ay.fl[k]);
    }
    c0 = clampi(c0, sm.S); c1 = clampi(c1 + 1, sm.S); r0 = clampi(r0, sm.S); r1 = clampi(r1 + 1, sm.S);
    if (c1 - c0 < kShadowTile && r1 - r0 < kShadowTile) {  // at most 2x2 tiles
        const int tw = shadow_tiles_w(sm.S);
        const float* T = sm.tiles(index);
        const int tc0 = c0 / kShadowTile, tc1 = c1 / kShadowTile, tr0 = r0 / kShadowTile, tr1 = r1 / kShadowTile;
        const float* a = T + 2 * ((size_t)tr0 * tw + tc0);
        const float* b = T + 2 * ((size_t)tr0 * tw + tc1);
        const float* c = T + 2 * ((size_t)tr1 * tw + tc0);
        const float* d = T + 2 * ((size_t)tr1 * tw + tc1);
        const float lo = fminf(fminf(a[0], b[0]), fminf(c[0], d[0])), hi = fmaxf(fmaxf(a[1], b[1]), fmaxf(c[1], d[1]));
        // every compare `ref <= texel` fails: each tap blends four zeros, lerpf(0, 0, f) = 0 + f * 0
        if (ref > hi) return 0.0f;
        // every compare passes: each tap is lerpf(1, 1, f) = 1 + f * 0 = 1 (the weights are finite)
        if (ref <= lo) {
            for (int k = 0; k < 4 * kShadowSamples * kShadowSamples; ++k) total += 1.0f * kInvShadowSamples;
            return total;
        }
    }
    bool lined_up = BLOCK;
    for (int k = 1; k < 4; ++k) lined_up = lined_up && ax.fl[k] == ax.fl[0] + k && ay.fl[k] == ay.fl[0] + k;
    if (lined_up) {
        const float* L = sm.data + (size_t)index * (size_t)sm.S * (size_t)sm.S;
        int col[5];
        size_t row[5];
        for (int m = 0; m < 5; ++m) {
            col[m] = clampi(ax.fl[0] + m, sm.S);
            row[m] = (size_t)clampi(ay.fl[0] + m, sm.S) * (size_t)sm.S;
        }
        uint32_t pass = 0u;  // bit 5 * r + c: ref <= texel (row r, column c) of the block
        for (int r = 0; r < 5; ++r)
            for (int c = 0; c < 5; ++c) pass |= (ref <= L[row[r] + col[c]] ? 1u : 0u) << (5 * r + c);
        for (int x = 0; x < 4; ++x)
            for (int y = 0; y < 4; ++y) {
                const uint32_t w = pass >> (5 * y + x);
                float c00 = (float)(w & 1u), c10 = (float)((w >> 1) & 1u), c01 = (float)((w >> 5) & 1u), c11 = (float)((w >> 6) & 1u);
                float s = lerpf(lerpf(c00, c10, ax.fr[x]), lerpf(c01, c11, ax.fr[x]), ay.fr[y]);
                total += s * kInvShadowSamples;
            }
        return total;
    }
    for (int x = -kShadowSamples; x < kShadowSamples; ++x)
        for (int y = -kShadowSamples; y < kShadowSamples; ++y) {
            float ox = (float)x / ts, oy = (float)y / ts;
            float s = shadow_compare(sm, lx + ox, ly + oy, index, ref);
            total += s * kInvShadowSamples;
        }
    return total;
}

// light.wgsl:31-61
template <bool BLOCK = true>
WR_HD float sample_direct_light(const FrameConsts& fc, const ShadowMap& sm, V4 wp) {
    float in_light = 0.0f;
    if (fc.use_shadowmaps > 0u) {
        // With matrix entries and position below 1e15 the light-space position is finite, so a
        // negative z gives ld.z = z * (1 / length) in [-inf, -0] (length > 0, possibly
        // overflowed to inf): below the first threshold, the face is skipped.  The three faces
        // that look away from the point take this exit before the rest of the product and the
        // normalisation are worked out.
        const float lim = 1.0e15f;
        const bool tame = fc.light_tame != 0u && fabsf(wp.x) < lim && fabsf(wp.y) < lim && fabsf(wp.z) < lim && fabsf(wp.w) < lim;
        for (int i = 0; i < 6; ++i) {
            const float* m = fc.light_m[i];
            const float lz = ((m[2] * wp.x + m[6] * wp.y) + m[10] * wp.z) + m[14] * wp.w;  // mat4_mul(m, wp).z
            if (tame && lz < 0.0f) continue;
            V4 lc = mat4_mul(fc.light_m[i], wp);
            V3 ld = normalize3(v3(lc.x, lc.y, lc.z));
            if (ld.z < kInvSqrt3 - 0.01f) continue;
            if (fabsf(ld.y) > kInvSqrt2 + 0.01f) continue;
            if (fabsf(ld.x) > kInvSqrt2 + 0.01f) continue;
            in_light = sample_direct_light_index<BLOCK>(sm, i, lc);
            if (in_light > 0.0f) break;
        }
    } else {
        in_light = 1.0f;
    }
    return in_light;
}

// light.wgsl:63-78
WR_HD V3 sample_ambient_light(const float* light, float light_dist, float surface_light_dot) {
    float att = 1.0f / ((1.0f + -0.75f * light_dist) + (0.25f * light_dist) * light_dist);
    V3 diff_direct = (v3(light[0], light[1], light[2]) * light[3]) * att;
    float diffuse_mult = fmaxf(surface_light_dot, 0.0f);
    return v3(0.01f, 0.01f, 0.01f) + diff_direct * (diffuse_mult * 0.03f);
}

// brdf.wgsl:35-65 with its helpers 3-33.  pow(x, 5.0) is evaluated as (x*x)*(x*x)*x.
WR_HD V3 brdf(V3 n, V3 l, V3 v, V3 h, V3 albedo, float roughness, float metalness) {
    V3 f0 = v3(mixf(0.04f, albedo.x, metalness), mixf(0.04f, albedo.y, metalness), mixf(0.04f, albedo.z, metalness));
    float b = 1.0f - fmaxf(dot3(h, v), 0.0f);
    float b2 = b * b;
    float p5 = (b2 * b2) * b;
    V3 fresnel = f0 + (v3(1.0f, 1.0f, 1.0f) - f0) * p5;
    float a2 = roughness * roughness;
    float ndh = fmaxf(dot3(n, h), 0.0f);
    float den = (ndh * ndh) * (a2 - 1.0f) + 1.0f;
    den = (kPi * den) * den;
    float ndf = a2 / den;
    float ndv = fmaxf(dot3(n, v), 0.0f);
    float ndl = fmaxf(dot3(n, l), 0.0f);
    float ggx1 = ndv / (ndv * (1.0f - roughness) + roughness);
    float ggx2 = ndl / (ndl * (1.0f - roughness) + roughness);
    float geo = ggx1 * ggx2;
    V3 nom = fresnel * (ndf * geo);
    float denom = (4.0f * ndv) * ndl + 0.0001f;
    V3 specular = nom / denom;
    V3 kd = (v3(1.0f, 1.0f, 1.0f) - fresnel) * (1.0f - metalness);
    return (((kd * albedo) / kPi) + specular) * ndl;
}

struct PbrInputs {
    V2 uv, ddx, ddy;
    V3 tpos, tlight, tview;
    V4 wpos;
};

// pbr.wgsl:73-130
WR_HD V4 fs_pbr(const FrameConsts& fc, const ShadowMap& sm, const uint32_t* texels, const TexDesc* tex,
                const MatDesc& m, const ColorTables& ct, const PbrInputs& in, V4 td) {  // td: the base-colour sample
    V4 tn = sample_material(texels, tex[m.normal], ct, in.uv, in.ddx, in.ddy);
    V4 trm = sample_material(texels, tex[m.mr], ct, in.uv, in.ddx, in.ddy);
    V3 albedo = v3(td.x, td.y, td.z);
    float roughness = trm.y * m.roughness;
    float metalness = trm.z * m.metallic;
    V3 total = v3(0.0f, 0.0f, 0.0f);
    V3 normal_dir = v3(tn.x * 2.0f - 1.0f, tn.y * 2.0f - 1.0f, tn.z * 2.0f - 1.0f);
    V3 light_dir = normalize3(in.tlight - in.tpos);
    float sld = dot3(normal_dir, light_dir);
    V3 lp = v3(fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]);
    float light_dist = length3(lp - v3(in.wpos.x, in.wpos.y, in.wpos.z));
    float att = 1.0f / (1.0f + light_dist * light_dist);
    V3 direct = (v3(fc.light_color[0], fc.light_color[1], fc.light_color[2]) * fc.light_color[3]) * att;
    float rs = fmaxf(sld, 0.0f);
    V3 view_dir = normalize3(in.tview - in.tpos);
    V3 half_dir = normalize3(view_dir + light_dir);
    // A surface turned away from the light has rs = 0, and the shader's radiance * brdf is
    // then (direct * 0) * in_light * brdf: a zero of either sign for any in_light as long as
    // direct and brdf are finite, and 0 + (+-0) leaves total_radiance at +0, exactly what an
    // unlit fragment gets.  The shadow lookup (six face tests, sixteen taps) is only made
    // where its result can reach the pixel.
    bool lookup = true;
    if (!(rs > 0.0f)) {
        V3 f = brdf(normal_dir, light_dir, view_dir, half_dir, albedo, roughness, metalness);
        const float big = 3.0e38f;
        lookup = !(fabsf(f.x) < big && fabsf(f.y) < big && fabsf(f.z) < big &&
                   fabsf(direct.x) < big && fabsf(direct.y) < big && fabsf(direct.z) < big);
    }
    float in_light = lookup ? sample_direct_light(fc, sm, in.wpos) : 0.0f;
    if (in_light > 0.0f) {
        V3 radiance = (direct * rs) * in_light;
        total = total + radiance * brdf(normal_dir, light_dir, view_dir, half_dir, albedo, roughness, metalness);
    }
    V3 ambient = sample_ambient_light(fc.light_color, light_dist, sld);
    ambient = ambient * albedo;
    V3 result = ambient + total;
    result = result / (result + v3(1.0f, 1.0f, 1.0f));
    return V4{result.x, result.y, result.z, td.w};
}

// ---------------------------------------------------------------------------------------
// Fog: noise.wgsl + fog.wgsl.

// fract(s) = s - floor(s).  The device's fraction instruction returns the same bits except
// where s - floor(s) rounds up to 1.0 (a negative s of less than 2^-25 in magnitude), which
// the noise hash cannot produce: its s is sin(d) * 42123.45 for the dot product d of an
// integer lattice point, d is 0 or at least 2^-20 in magnitude, and for every float d with
// 2^-20 <= |d| < 2^30 the sine is at least 8.3e-9 in magnitude, so the difference stays below 1
// (checked exhaustively: wro_hash_fract_exceptions in the oracle, run by the CPU tests).
WR_HD float hash_fract(float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(s);
#else
    return s - floorf(s);
#endif
}
// noise.wgsl:4 after the dot product: fract(sin(d) * 42123.45) * 2 - 1.  fr * 2 is exact,
// so the fused form rounds once exactly where the two-step form rounds.
template <int GUARD>
WR_HD float hash_from_dot(float d) {
    float s = wr_sin_t<GUARD>(d) * 42123.45f;
    return fmaf(hash_fract(s), 2.0f, -1.0f);
}
// smoothstep(0, 1, v) = t*t*(3 - 2t); 2t is exact, so fma(-2, t, 3) equals 3 - 2*t bit for bit.
WR_HD float smooth01(float v) {
    float t = fminf(fmaxf(v, 0.0f), 1.0f);
    return (t * t) * fmaf(-2.0f, t, 3.0f);
}
// noise.wgsl:7-24.  The eight corner dot products (x*12.345 + y*67.89) + z*412.12 share
// their six products and four x+y sums; every value is rounded exactly as when each
// corner is evaluated on its own.
template <int GUARD>
WR_HD float value_noise_t(V3 p) {
    float ux = floorf(p.x), uy = floorf(p.y), uz = floorf(p.z);
    float sx = smooth01(p.x - ux), sy = smooth01(p.y - uy), sz = smooth01(p.z - uz);
    float x0 = ux * 12.345f, x1 = (ux + 1.0f) * 12.345f;
    float y0 = uy * 67.89f, y1 = (uy + 1.0f) * 67.89f;
    float z0 = uz * 412.12f, z1 = (uz + 1.0f) * 412.12f;
    float xy00 = x0 + y0, xy10 = x1 + y0, xy01 = x0 + y1, xy11 = x1 + y1;
    float a = hash_from_dot<GUARD>(xy00 + z0), b = hash_from_dot<GUARD>(xy10 + z0);
    float c = hash_from_dot<GUARD>(xy01 + z0), d = hash_from_dot<GUARD>(xy11 + z0);
    float e = hash_from_dot<GUARD>(xy00 + z1), f = hash_from_dot<GUARD>(xy10 + z1);
    float g = hash_from_dot<GUARD>(xy01 + z1), h = hash_from_dot<GUARD>(xy11 + z1);
    return mixf(mixf(mixf(a, b, sx), mixf(c, d, sx), sy), mixf(mixf(e, f, sx), mixf(g, h, sx), sy), sz);
}
WR_HD float value_noise(V3 p) { return value_noise_t<true>(p); }

// Per-frame hash tables for the low-frequency octaves.  The hash of a lattice point depends
// only on its integer coordinates, so for an octave whose lattice over the fog region is
// small it is computed once per frame (k_fog_tables) and the march reads the eight corner
// hashes instead of evaluating eight sines.  An entry holds exactly hash_from_dot of the
// spec's dot product for that lattice point, so a table hit and a direct evaluation give
// the same bits; a sample outside the table's box falls back to the direct evaluation.
struct NoiseTable {
    const float* data;   // nullptr: level not tabled
    // What a lookup reads, derived from the box below by noise_table_seal.  Lattice coordinates
    // here are integers below 2^24 in magnitude, so they, their differences and an index into
    // at most kNoiseTableMaxEntries entries are exact in f32.
    float flo[3];        // lox, loy, loz
    uint32_t lim[3];     // bits of (float)(n - 1): u - lo lies in [0, n - 1) iff its bits, as an unsigned integer, are below
    float fnx, fny;
    uint32_t row_bytes, slab_bytes;  // 4 * nx, 4 * nx * ny
    // fbm_decided's bounds for this level's octave (set for every level, tabled or not, by
    // noise_tables_clear): a partial sum at or below the first or at or above the second decides.
    // They ride with the description, which the march fetches anyway with scalar loads.
    float decided_low, decided_high;
    int lox, loy, loz;   // lattice coordinate of entry (0,0,0)
    int nx, ny, nz;
#ifdef WR_DEV
    int dbg_level, dbg_pad;  // level index, + 256 in the light-march kernel
#endif
};
constexpr uint32_t kNoiseTableMaxEntries = 1u << 24;
struct NoiseTables { NoiseTable level[16]; };  // [fbm index * 8 + octave]

WR_HD float lattice_hash(float ux, float uy, float uz) {  // rand(u) of noise.wgsl:3-5 for integer u
    return hash_from_dot<true>((ux * 12.345f + uy * 67.89f) + uz * 412.12f);
}

#if defined(WR_DEV)
// Developer build: lane counts per noise level, [kernel kind][level][0 = table hit, 1 = direct].
__device__ unsigned long long g_dbg_levels[2][16][2];
#endif
#if defined(WR_DEV)
// Developer build: directly evaluated octaves (lanes) by the size of the lattice box their
// wavefront's direct lanes span, [kernel kind][level][bucket: <= 64, 128, 256, 512, 1024, more points].
__device__ unsigned long long g_dbg_boxes[2][16][6];
#endif
#if defined(WR_DEV) && defined(__HIP_DEVICE_COMPILE__)
#define WR_DBG_LEVEL(level, direct) atomicAdd(&g_dbg_levels[(level) >> 8][(level) & 15][direct], 1ull)
__device__ inline void wr_dbg_box(int level, float ux, float uy, float uz) {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    const unsigned long long lanes = __ballot(1);
    for (unsigned long long m = lanes; m; m &= m - 1ull) {
        const int l = (int)__builtin_ctzll(m);
        const int c[3] = {__builtin_amdgcn_readlane((int)ux, l), __builtin_amdgcn_readlane((int)uy, l), __builtin_amdgcn_readlane((int)uz, l)};
        for (int k = 0; k < 3; ++k) { lo[k] = c[k] < lo[k] ? c[k] : lo[k]; hi[k] = c[k] > hi[k] ? c[k] : hi[k]; }
    }
    const long long n = (long long)(hi[0] - lo[0] + 2) * (hi[1] - lo[1] + 2) * (hi[2] - lo[2] + 2);
    const int bucket = n <= 64 ? 0 : n <= 128 ? 1 : n <= 256 ? 2 : n <= 512 ? 3 : n <= 1024 ? 4 : 5;
    if ((int)__builtin_ctzll(lanes) == (int)(threadIdx.x & 63u))
        atomicAdd(&g_dbg_boxes[level >> 8][level & 15][bucket], (unsigned long long)__builtin_popcountll(lanes));
}
#else
#define WR_DBG_LEVEL(level, direct) ((void)0)
#endif
template <int GUARD>
WR_HD float value_noise_tab(V3 p, const NoiseTable& t) {
    float ux = floorf(p.x), uy = floorf(p.y), uz = floorf(p.z);
    // The level's description, fetched in one piece (one scalar load and one wait on the device,
    // where a field fetched inside each nested test was a wait per field).
    NoiseTable d = t;
#if defined(__HIP_DEVICE_COMPILE__)
    // (every field now, in scalar registers: left alone, the compiler fetches the fields of the
    // box behind the test that follows and waits a second time.  Not the pointer: one that has
    // been through an asm statement is no longer known to be global memory.)
    asm volatile("" : "+s"(d.flo[0]), "+s"(d.flo[1]), "+s"(d.flo[2]), "+s"(d.lim[0]), "+s"(d.lim[1]), "+s"(d.lim[2]),
                      "+s"(d.fnx), "+s"(d.fny), "+s"(d.row_bytes), "+s"(d.slab_bytes));
#endif
    if (d.data != nullptr) {  // (the same in every lane: a level without a table skips the index arithmetic)
        // position in the box, exact; a negative difference has its sign bit set and fails the
        // unsigned comparison like one that is too large.  (-0 - 0 = -0 fails it too: the direct
        // evaluation below gives the same bits.)
        const float fx = ux - d.flo[0], fy = uy - d.flo[1], fz = uz - d.flo[2];
        const bool inside = ((int)(f2u(fx) < d.lim[0]) & (int)(f2u(fy) < d.lim[1]) & (int)(f2u(fz) < d.lim[2])) != 0;  // (no short circuit: one branch)
        if (inside) {
            float sx = smooth01(p.x - ux), sy = smooth01(p.y - uy), sz = smooth01(p.z - uz);
            // (z * ny + y) * nx + x: every product and sum an integer below 2^24, so exact, fused or not
            const uint32_t o00 = (uint32_t)fmaf(fmaf(fz, d.fny, fy), d.fnx, fx) << 2;
            const char* base = reinterpret_cast<const char*>(d.data);
            const float* q00 = reinterpret_cast<const float*>(base + o00);
            const float* q01 = reinterpret_cast<const float*>(base + (o00 + d.row_bytes));
            const float* q10 = reinterpret_cast<const float*>(base + (o00 + d.slab_bytes));
            const float* q11 = reinterpret_cast<const float*>(base + (o00 + d.slab_bytes + d.row_bytes));
            float a = q00[0], b = q00[1], c = q01[0], d_ = q01[1];
            float e = q10[0], f = q10[1], g = q11[0], h = q11[1];
#ifdef WR_DEV
            WR_DBG_LEVEL(d.dbg_level, 0);
#endif
            return mixf(mixf(mixf(a, b, sx), mixf(c, d_, sx), sy), mixf(mixf(e, f, sx), mixf(g, h, sx), sy), sz);
        }
    }
#ifdef WR_DEV
    WR_DBG_LEVEL(t.dbg_level, 1);
#if defined(__HIP_DEVICE_COMPILE__)
    wr_dbg_box(t.dbg_level, ux, uy, uz);
#endif
#endif
    return value_noise_t<GUARD>(p);
}

// fbm returns saturate(sum); once the partial sum is so far below 0 (or above 1) that the
// octaves still to come cannot bring it back, the result is known: 0 (or 1), the same bits
// the full sum would saturate to.  What is still to come after octave i is at most
// S_i = 2^-(i+1) - 2^-8 in magnitude times 1 + 4e-7 (a value_noise is three levels of mixes
// of hashes in [-1, 1], each mix rounding three times) plus seven roundings of a sum of about
// 1 (4.2e-7): less than S_i + 1e-6, and the test asks for 2e-6.  On the device the octaves
// stop only when every lane of the wavefront is decided (lanes of one wavefront are
// neighbouring pixels, and the low octaves that decide vary slowly in space); a decided lane
// that runs on still saturates to the same value.  NaN decides nothing.
#define WR_FBM_REST(i) ((0.5f / (float)(1 << (i))) - 0.00390625f + 2.0e-6f)  // S_i + 2e-6
WR_HD bool fbm_decided_between(float ret, float low, float high) {
#if defined(__HIP_DEVICE_COMPILE__)
    // every active lane decided, from the lane masks of the two comparisons (a vote on their
    // disjunction is compiled to two more vector instructions)
    return (__builtin_amdgcn_ballot_w64(ret <= low) | __builtin_amdgcn_ballot_w64(ret >= high)) == __builtin_amdgcn_ballot_w64(true);
#else
    return ret <= low || ret >= high;
#endif
}
WR_HD bool fbm_decided(float ret, int i) {
    const float rest = WR_FBM_REST(i);
    return fbm_decided_between(ret, -rest, 1.0f + rest);
}
// The lookup's fields of a level whose box (lo, n) has been set.
WR_HD void noise_table_seal(NoiseTable& t) {
    t.flo[0] = (float)t.lox; t.flo[1] = (float)t.loy; t.flo[2] = (float)t.loz;
    t.lim[0] = t.nx > 1 ? f2u((float)(t.nx - 1)) : 0u;
    t.lim[1] = t.ny > 1 ? f2u((float)(t.ny - 1)) : 0u;
    t.lim[2] = t.nz > 1 ? f2u((float)(t.nz - 1)) : 0u;
    t.fnx = (float)t.nx; t.fny = (float)t.ny;
    t.row_bytes = 4u * (uint32_t)t.nx;
    t.slab_bytes = 4u * (uint32_t)t.nx * (uint32_t)t.ny;
}
// No level tabled; every level's bounds as fbm_decided computes them.
WR_HD void noise_tables_clear(NoiseTables& nt) {
    for (int l = 0; l < 16; ++l) {
        NoiseTable& t = nt.level[l];
        t.data = nullptr;
        t.lox = t.loy = t.loz = 0;
        t.nx = t.ny = t.nz = 0;
        noise_table_seal(t);
        const float rest = WR_FBM_REST(l & 7);
        t.decided_low = -rest;
        t.decided_high = 1.0f + rest;
#ifdef WR_DEV
        t.dbg_level = l; t.dbg_pad = 0;
#endif
    }
}
template <int GUARD>
WR_HD float fbm_t(V3 p) {
    float weight = 0.5f, ret = 0.0f;
    V3 q = p;
    for (int i = 0; i < 8; ++i) {
        ret += weight * value_noise_t<GUARD>(q);
        if (i < 7 && fbm_decided(ret, i)) break;
        q = q * 2.0f;
        weight *= 0.5f;
    }
    return saturatef(ret);
}
template <int GUARD>
WR_HD float fbm_tab_t(V3 p, const NoiseTable* tabs) {
    float weight = 0.5f, ret = 0.0f;
    V3 q = p;
    for (int i = 0; i < 8; ++i) {
        const NoiseTable& t = tabs[i];
        float low = t.decided_low, high = t.decided_high;
#if defined(__HIP_DEVICE_COMPILE__)
        // (the level index is the same in every lane: the two bounds are fetched here, next to
        // the level's description, and not behind the noise, where their fetch would be waited for)
        asm volatile("" : "+s"(low), "+s"(high));
#endif
        ret += weight * value_noise_tab<GUARD>(q, t);
        if (i < 7 && fbm_decided_between(ret, low, high)) break;
        q = q * 2.0f;
        weight *= 0.5f;
    }
    return saturatef(ret);
}
// noise.wgsl:26-40.  The last octave samples p*128 and every hash argument is at most
// (|q|+1) * (12.345 + 67.89 + 412.12) (1 + 3*2^-24); for |p| < 16000 that stays below 2^30,
// the sine's range guard can never fire and an unguarded instantiation is used; for
// |p| < 200 it stays below kSinFastRange (12604801 < 13176794) and the fast form serves.  The
// first bound also keeps the float -> int conversions that index the hash tables exact
// (|p| * 128 + 1 < 2^24).
constexpr float kFbmUnguarded = 16000.0f, kFbmFast = 200.0f;
WR_HD int fbm_sin_tier(V3 p) {  // NaN falls through to the guarded form
#ifndef WR_DEV_NO_FAST_SINE
    if (fabsf(p.x) < kFbmFast && fabsf(p.y) < kFbmFast && fabsf(p.z) < kFbmFast) return kSinFast;
#endif
    if (fabsf(p.x) < kFbmUnguarded && fabsf(p.y) < kFbmUnguarded && fabsf(p.z) < kFbmUnguarded) return kSinInt;
    return kSinGuarded;
}
WR_HD float fbm(V3 p) {
    const int tier = fbm_sin_tier(p);
    if (tier == kSinFast) return fbm_t<kSinFast>(p);
    if (tier == kSinInt) return fbm_t<kSinInt>(p);
    return fbm_t<kSinGuarded>(p);
}
WR_HD float fbm_tab(V3 p, const NoiseTable* tabs) {
    const int tier = fbm_sin_tier(p);
    if (tier == kSinFast) return fbm_tab_t<kSinFast>(p, tabs);
    if (tier == kSinInt) return fbm_tab_t<kSinInt>(p, tabs);
    return fbm_t<kSinGuarded>(p);
}
// fog.wgsl:48-62
WR_HD float fog_noise(const FrameConsts& fc, V3 pos) {
    V3 p1 = pos * 0.01f;
    p1.x += fc.drift_xy; p1.y += fc.drift_xy; p1.z += fc.drift_z;
    float n1 = fbm(p1);
    V3 p2 = pos * 0.05f;
    p2.x += fc.drift_xy; p2.y += fc.drift_xy; p2.z += fc.drift_z;
    float n2 = fbm(p2);
    return (0.8f * n1 + 0.15f * n2) + 0.05f;
}
WR_HD float fog_noise_tab(const FrameConsts& fc, const NoiseTables& nt, V3 pos) {
    V3 p1 = pos * 0.01f;
    p1.x += fc.drift_xy; p1.y += fc.drift_xy; p1.z += fc.drift_z;
    float n1 = fbm_tab(p1, nt.level);
    V3 p2 = pos * 0.05f;
    p2.x += fc.drift_xy; p2.y += fc.drift_xy; p2.z += fc.drift_z;
    float n2 = fbm_tab(p2, nt.level + 8);
    return (0.8f * n1 + 0.15f * n2) + 0.05f;
}
// fog.wgsl:64-85.  Returns (density, depth); `evals` counts fog_noise calls.  nt == nullptr
// evaluates every hash directly.
WR_HD V2 ray_march(const FrameConsts& fc, const NoiseTables* nt, V3 origin, V3 dir, float max_depth, int max_steps,
                   float step_size, float fog_density, uint32_t& evals) {
    float density = 0.0f, depth = 0.0f;
    float contribution = fog_density / (float)max_steps;
    for (int i = 0; i < max_steps; ++i) {
        depth += step_size;
        if (depth >= max_depth) break;
        V3 pos = origin + dir * depth;
        float noise = nt ? fog_noise_tab(fc, *nt, pos) : fog_noise(fc, pos);
        ++evals;
        density += noise * contribution;
        if (density >= 1.0f) { density = 1.0f; break; }
    }
    return V2{density, depth};
}
// fog.wgsl:118-123
WR_HD float depth_to_linear(float depth, float zn, float zf) { return (zn * zf) / (zf + depth * (zn - zf)); }

// fog.wgsl:125-136: the part of fs_main before the march.  Returns max_fog_depth; a value
// <= 0 means the shader returns vec4(0), which blends to an unchanged destination.
// depth_tex is the geometry depth texture (sampler texture.rs:59-72: ClampToEdge, Linear,
// no compare).
WR_HD float fog_max_depth(const FrameConsts& fc, const float* depth_tex, float frag_z, int px, int py) {
    float volume_depth = depth_to_linear(frag_z, fc.planes[0], fc.planes[1]);
    float u = ((float)px + 0.5f) / fc.planes[2], v = ((float)py + 0.5f) / fc.planes[3];
    float x = guard_coord(u * (float)fc.W - 0.5f), y = guard_coord(v * (float)fc.H - 0.5f);
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int i0 = clampi((int)x0, fc.W), i1 = clampi((int)x0 + 1, fc.W);
    int j0 = clampi((int)y0, fc.H), j1 = clampi((int)y0 + 1, fc.H);
    float t00 = depth_tex[(size_t)j0 * fc.W + i0], t10 = depth_tex[(size_t)j0 * fc.W + i1];
    float t01 = depth_tex[(size_t)j1 * fc.W + i0], t11 = depth_tex[(size_t)j1 * fc.W + i1];
    float gd = lerpf(lerpf(t00, t10, fx), lerpf(t01, t11, fx), fy);
    float geometry_depth = depth_to_linear(gd, fc.planes[0], fc.planes[1]);
    return geometry_depth - volume_depth;
}

// ray_march with the noise of its steps evaluated beforehand (noise_at(i) returns what
// fog_noise gives at origin + dir * depth_i): the same loop, the same additions, the same exits.
template <class NoiseAt>
WR_HD V2 ray_march_stored(float max_depth, int max_steps, float step_size, float fog_density, NoiseAt noise_at, uint32_t& evals) {
    float density = 0.0f, depth = 0.0f;
    float contribution = fog_density / (float)max_steps;
#pragma unroll
    for (int i = 0; i < max_steps; ++i) {
        depth += step_size;
        if (depth >= max_depth) break;
        float noise = noise_at(i);
        ++evals;
        density += noise * contribution;
        if (density >= 1.0f) { density = 1.0f; break; }
    }
    return V2{density, depth};
}
// The stored noise of a march, fetched in one go: every step the depth exit allows has a value
// (the density exit was not applied when they were written), so the loads do not depend on
// the running sum and can all be in flight together; ray_march_stored then reads registers.
template <int STEPS>
struct StoredNoise {
    float v[STEPS];
    WR_HD void fetch(const float* column, size_t stride, float step_size, float max_depth) {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) v[i] = step_size * (float)(i + 1) < max_depth ? column[(size_t)i * stride] : 0.0f;
    }
};
// Value of `depth` in step i of ray_march (i from 0): step_size added i + 1 times to 0.  The
// step sizes of constants.wgsl are the integers 5 and 10 and there are at most 30 steps, so
// every partial sum is an integer below 2^24 and the product below is that sum exactly.
WR_HD float march_step_depth(float step_size, int i) { return step_size * (float)(i + 1); }
static_assert(kFogStepSize == 5.0f && kFogLightStepSize == 10.0f && kFogMaxSteps <= 30 && kFogLightMaxSteps <= 30,
              "march_step_depth relies on integer step sizes");

// Where the light march of fog.wgsl:150-159 starts and which way it goes.
struct LightRay { V3 origin, dir; float max_dist; };
WR_HD LightRay fog_light_ray(const FrameConsts& fc, V3 fog_end) {
    V3 lp = v3(fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]);
    V3 to_light = lp - fog_end;
    float max_light_dist = length3(to_light);
    return LightRay{fog_end, to_light / max_light_dist, max_light_dist};
}
// fog.wgsl:138-165 after both marches: the fog colour.  `occlusion` is read only for a lit
// fog point (in_light > 0), as in the shader.
WR_HD V4 fog_shade(const FrameConsts& fc, V3 fog_position, float fog_density, float in_light, float occlusion) {
    V3 lp = v3(fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]);
    float light_dist = length3(lp - fog_position);
    float bc = mixf(kFogDensityColorX, kFogDensityColorY, fog_density);
    V3 ambient = sample_ambient_light(fc.light_color, light_dist, 1.0f) * 2.0f;
    ambient = ambient * v3(bc, bc, bc);
    V3 radiance = v3(0.0f, 0.0f, 0.0f);
    if (in_light > 0.0f) {
        float att = 1.0f / (1.0f + light_dist * light_dist);
        radiance = (((v3(fc.light_color[0], fc.light_color[1], fc.light_color[2]) * fc.light_color[3]) * att) * in_light) * (1.0f - occlusion);
    }
    V3 result = ambient + radiance;
    result = result / (result + v3(1.0f, 1.0f, 1.0f));
    return V4{result.x, result.y, result.z, fog_density * kFogAlpha};
}

// fog.wgsl:127-128,138-165 (ray_march_fog 87-116 inlined): the march and the fog colour for
// a fragment whose max_fog_depth is positive.
WR_HD V4 fog_march_shade(const FrameConsts& fc, const ShadowMap& sm, const NoiseTables* nt, V3 origin, float max_fog_depth,
                         uint32_t& evals) {
    V3 cam = v3(fc.cam_pos[0], fc.cam_pos[1], fc.cam_pos[2]);
    V3 direction = normalize3(origin - cam);
    V2 fm = ray_march(fc, nt, origin, direction, max_fog_depth, kFogMaxSteps, kFogStepSize, kFogDensity, evals);
    float fog_density = fm.x, fog_depth = fm.y;
    V3 fog_position = origin + direction * fog_depth;  // fog_end and fog_position of the shader are this one point
    float in_light = sample_direct_light<false>(fc, sm, V4{fog_position.x, fog_position.y, fog_position.z, 1.0f});
    float occlusion = 0.0f;
    if (in_light > 0.0f) {
        // The light march (fog.wgsl:150-159) only feeds `occlusion`, which only the lit branch
        // reads: a fog point in shadow does not need it, and it is not run for one.
        LightRay lr = fog_light_ray(fc, fog_position);
        V2 lm = ray_march(fc, nt, lr.origin, lr.dir, lr.max_dist, kFogLightMaxSteps, kFogLightStepSize, kFogLightDensity, evals);
        occlusion = lm.x;
    }
    return fog_shade(fc, fog_position, fog_density, in_light, occlusion);
}

}  // namespace wr
