// Host-side unit access to the product's __host__ __device__ arithmetic (csrc/wr_device.h),
// so that CPU-only tests can compare it bit for bit with the oracle before any GPU run.
// Test infrastructure: exposes single functions only, no rendering path.
#include <cstring>
#include <cmath>
#include <hip/hip_runtime.h>
#include "../../wgpu-renderer_amd/csrc/wr_device.h"
using namespace wr;

// Device-side unit checks (GPU tests only): the same functions evaluated by a kernel.
__global__ void k_hc_hash(const float* d, int n, float* out_guarded, float* out_unguarded) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_guarded[i] = hash_from_dot<kSinGuarded>(d[i]);
    // the unguarded forms, each where its caller may use it (elsewhere the guarded value is passed on)
    const float ad = fabsf(d[i]);
    out_unguarded[i] = ad <= kSinFastRange ? hash_from_dot<kSinFast>(d[i]) : hash_from_dot<kSinInt>(d[i]);
}
__global__ void k_hc_sin(const float* x, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = wr_sin(x[i]);
}
// Over every f32 bit pattern: how often the hardware fraction differs from s - floor(s) where
// the latter is a number below 1 (where it rounds up to 1 the hardware returns the float below).
__global__ void k_hc_fract(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32);
         b += (unsigned long long)gridDim.x * blockDim.x) {
        float s = u2f((uint32_t)b);
        float want = s - floorf(s);
        if (!(want < 1.0f)) continue;  // also NaN and infinities
        if (f2u(__builtin_amdgcn_fractf(s)) != f2u(want)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

extern "C" {
// returns 0, or the HIP error code
int hc_dev_hash(const float* d, int n, float* out_guarded, float* out_unguarded) {
    float *dd = nullptr, *dg = nullptr, *du = nullptr;
    hipError_t e = hipMalloc(&dd, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dg, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&du, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(dd, d, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_hc_hash<<<(n + 255) / 256, 256>>>(dd, n, dg, du);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(out_guarded, dg, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_unguarded, du, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dd); (void)hipFree(dg); (void)hipFree(du);
    return (int)e;
}
int hc_dev_sin(const float* x, int n, float* out) {
    float *dx = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&dx, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_hc_sin<<<(n + 255) / 256, 256>>>(dx, n, dout);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dx); (void)hipFree(dout);
    return (int)e;
}
long long hc_dev_fract_mismatches() {
    unsigned long long* dm = nullptr;
    unsigned long long m = 0;
    if (hipMalloc(&dm, 8) != hipSuccess) return -1;
    if (hipMemset(dm, 0, 8) != hipSuccess) return -1;
    k_hc_fract<<<4096, 256>>>(dm);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(&m, dm, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipFree(dm);
    return (long long)m;
}
float hc_hash_from_dot(float d) { return hash_from_dot<true>(d); }
// Host build of the product's sine: floats with bit patterns lo_bits + phase, + stride, ... below
// hi_bits (both signs) for which the fast form and the guarded form differ.
unsigned long long hc_sin_fast_mismatches(unsigned lo_bits, unsigned hi_bits, unsigned stride, unsigned phase) {
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)lo_bits + phase; b < hi_bits; b += stride)
        for (unsigned sg = 0; sg < 2; ++sg) {
            const float x = u2f((unsigned)b | (sg << 31));
            if (f2u(wr_sin_t<kSinFast>(x)) != f2u(wr_sin_t<kSinGuarded>(x))) ++bad;
            if (f2u(wr_sin_t<kSinInt>(x)) != f2u(wr_sin_t<kSinGuarded>(x))) ++bad;
        }
    return bad;
}
float hc_sin_fast_range() { return kSinFastRange; }
float hc_sin(float x) { return wr_sin(x); }
float hc_value_noise(const float* p) { return value_noise(v3(p[0], p[1], p[2])); }
float hc_fbm(const float* p) { return fbm(v3(p[0], p[1], p[2])); }
float hc_fog_noise(const float* p, float time) {
    FrameConsts fc; memset(&fc, 0, sizeof(fc));
    fc.time = time; fc.drift_xy = time * 0.2f; fc.drift_z = wr_sin(time * 0.1f) * 0.5f;
    return fog_noise(fc, v3(p[0], p[1], p[2]));
}
// fog_noise through hash tables built on the host for a box around p (every level whose
// lattice box has at most `cap` entries), to compare with the direct evaluation.
float hc_fog_noise_tab(const float* p, float time, int cap, int* tabled_levels) {
    FrameConsts fc; memset(&fc, 0, sizeof(fc));
    fc.time = time; fc.drift_xy = time * 0.2f; fc.drift_z = wr_sin(time * 0.1f) * 0.5f;
    NoiseTables nt; memset(&nt, 0, sizeof(nt));
    static float pool[1 << 16];
    size_t used = 0; int nlev = 0;
    const double drift[3] = {fc.drift_xy, fc.drift_xy, fc.drift_z};
    for (int k = 0; k < 2; ++k) for (int o = 0; o < 8; ++o) {
        double s = k == 0 ? 0.01 : 0.05, m = (double)(1 << o);
        long l[3], n[3];
        for (int r = 0; r < 3; ++r) {
            double a = ((p[r] - 3.0) * s + drift[r]) * m, b = ((p[r] + 3.0) * s + drift[r]) * m;
            l[r] = (long)floor(a) - 1; n[r] = (long)floor(b) + 2 - l[r] + 1;
        }
        size_t entries = (size_t)(n[0] * n[1] * n[2]);
        if ((int)entries > cap || used + entries > sizeof(pool) / 4) continue;
        NoiseTable& t = nt.level[k * 8 + o];
        t.data = pool + used; t.lox = (int)l[0]; t.loy = (int)l[1]; t.loz = (int)l[2];
        t.nx = (int)n[0]; t.ny = (int)n[1]; t.nz = (int)n[2];
        for (size_t i = 0; i < entries; ++i) {
            int ix = (int)(i % n[0]), iy = (int)((i / n[0]) % n[1]), iz = (int)(i / (n[0] * n[1]));
            pool[used + i] = lattice_hash((float)(t.lox + ix), (float)(t.loy + iy), (float)(t.loz + iz));
        }
        used += entries; ++nlev;
    }
    *tabled_levels = nlev;
    return fog_noise_tab(fc, nt, v3(p[0], p[1], p[2]));
}
void hc_brdf(const float* n, const float* l, const float* v, const float* h, const float* a, float rough, float metal, float* out) {
    V3 r = brdf(v3(n[0], n[1], n[2]), v3(l[0], l[1], l[2]), v3(v[0], v[1], v[2]), v3(h[0], h[1], h[2]), v3(a[0], a[1], a[2]), rough, metal);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void hc_ambient(const float* light, float dist, float sld, float* out) {
    V3 r = sample_ambient_light(light, dist, sld);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float hc_depth_to_linear(float d, float zn, float zf) { return depth_to_linear(d, zn, zf); }
static float g_tables[768];
static bool g_tables_ok = false;
static void tables() {
    if (g_tables_ok) return;
    auto to_linear = [](double s) { return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4); };
    for (int c = 0; c < 256; ++c) {
        g_tables[c] = (float)to_linear((double)c / 255.0);
        g_tables[256 + c] = (float)c / 255.0f;
        g_tables[512 + c] = c == 0 ? 0.0f : (float)to_linear(((double)c - 0.5) / 255.0);
    }
    g_tables_ok = true;
}
unsigned hc_srgb_encode(float x) { tables(); ColorTables ct{g_tables}; return srgb_encode(ct, x); }
unsigned hc_blend_pack(unsigned dst, const float* src, int blend) { tables(); ColorTables ct{g_tables}; return blend_pack(ct, dst, V4{src[0], src[1], src[2], src[3]}, blend != 0); }

// Same 28-word-per-sub-triangle layout as the oracle's wro_setup_triangle.
int hc_setup_triangle(const float* clip, unsigned W, unsigned H, int shadow, unsigned* out, int max_subs) {
    V4 c[3];
    for (int k = 0; k < 3; ++k) c[k] = V4{clip[4 * k], clip[4 * k + 1], clip[4 * k + 2], clip[4 * k + 3]};
    int cls = classify_triangle(c);
    if (cls == 0) return 0;
    ClipVert poly[10];
    for (int k = 0; k < 3; ++k) {
        poly[k].x = c[k].x; poly[k].y = c[k].y; poly[k].z = c[k].z; poly[k].w = c[k].w;
        poly[k].b0 = k == 0; poly[k].b1 = k == 1; poly[k].b2 = k == 2;
    }
    int n = 3;
    if (cls == 2) { n = clip_polygon(poly, 3); if (n < 3) return 0; }
    int cnt = 0;
    for (int i = 1; i + 1 < n && cnt < max_subs; ++i) {
        ClipVert cv[3] = {poly[0], poly[i], poly[i + 1]};
        Rec r; float beta[9];
        if (!setup_sub(cv, (int)W, (int)H, shadow != 0, (unsigned)(i - 1), r, beta)) continue;
        unsigned* w = out + (size_t)cnt * 28;
        w[0] = r.X0; w[1] = r.Y0; w[2] = r.X1; w[3] = r.Y1; w[4] = r.X2; w[5] = r.Y2;
        w[6] = f2u(r.z0); w[7] = f2u(r.dz1); w[8] = f2u(r.dz2); w[9] = f2u(r.inv2A);
        if (shadow) {  // bias lives in iw0 for shadow records
            w[10] = 0; w[11] = f2u(r.iw1); w[12] = f2u(r.iw2); w[13] = f2u(r.iw0);
        } else {
            w[10] = f2u(r.iw0); w[11] = f2u(r.iw1); w[12] = f2u(r.iw2); w[13] = 0;
        }
        w[14] = r.id & 7u; w[15] = r.bbmin & 0xffffu; w[16] = r.bbmin >> 16; w[17] = r.bbmax & 0xffffu; w[18] = r.bbmax >> 16;
        for (int k = 0; k < 9; ++k) w[19 + k] = f2u(beta[k]);
        ++cnt;
    }
    return cnt;
}
// coverage + depth + barycentrics of one pixel for the first sub-triangle: returns 1 if covered
int hc_fragment(const float* clip, unsigned W, unsigned H, int px, int py, float* out_z, float* out_B) {
    V4 c[3];
    for (int k = 0; k < 3; ++k) c[k] = V4{clip[4 * k], clip[4 * k + 1], clip[4 * k + 2], clip[4 * k + 3]};
    if (classify_triangle(c) != 1) return -1;
    ClipVert cv[3];
    for (int k = 0; k < 3; ++k) { cv[k].x = c[k].x; cv[k].y = c[k].y; cv[k].z = c[k].z; cv[k].w = c[k].w; cv[k].b0 = cv[k].b1 = cv[k].b2 = 0; }
    Rec r;
    if (!setup_sub(cv, (int)W, (int)H, false, 0, r, nullptr)) return -1;
    Edges e; make_edges(r, e);
    int64_t E[3]; edge_eval(e, px, py, E);
    if (!edges_inside(e, E)) return 0;
    *out_z = depth_at(r, E);
    bary_at(r, nullptr, E, out_B);
    return 1;
}
}
