// Host-side unit access to the product's __host__ __device__ arithmetic (csrc/wr_device.h),
// so that CPU-only tests can compare it bit for bit with the oracle before any GPU run.
// Test infrastructure: exposes single functions only, no rendering path.
#include <cstring>
#include <cmath>
#include <hip/hip_runtime.h>
#include "../../wgpu-renderer_amd/csrc/wr_device.h"
using namespace wr;

// Same 28-word-per-sub-triangle layout as the oracle's wro_setup_triangle.
WR_HD int setup_triangle_words(const float* clip, unsigned W, unsigned H, int shadow, unsigned* out, int max_subs) {
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


// The rest of the arithmetic, evaluated by kernels over arrays of inputs (GPU tests compare the
// results with the oracle's for the same inputs).  One item per thread, 256 threads per group,
// so that neighbouring lanes of a wavefront hold unrelated items: the worst case for whatever
// a function decides per wavefront (the early end of fbm's octave loop).
__global__ void k_hc_noise(const float* p, int n, float time, float drift_z, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FrameConsts fc; memset(&fc, 0, sizeof(fc));
    fc.time = time; fc.drift_xy = time * 0.2f; fc.drift_z = drift_z;
    const V3 q = v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    out[3 * i] = value_noise(q);
    out[3 * i + 1] = fbm(q * 0.05f);
    out[3 * i + 2] = fog_noise(fc, q);
}
__global__ void k_hc_noise_tab(const float* p, int n, FrameConsts fc, NoiseTables nt, float* out_tab, float* out_direct) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 q = v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    out_tab[i] = fog_noise_tab(fc, nt, q);
    out_direct[i] = fog_noise(fc, q);
}
// 17 floats in (n, l, v, h, albedo, roughness, metalness), 3 out
__global__ void k_hc_brdf(const float* in, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = in + 17 * (size_t)i;
    V3 r = brdf(v3(b[0], b[1], b[2]), v3(b[3], b[4], b[5]), v3(b[6], b[7], b[8]), v3(b[9], b[10], b[11]), v3(b[12], b[13], b[14]), b[15], b[16]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
// 6 floats in (light colour rgba, distance, surface-light dot), 3 out
__global__ void k_hc_ambient(const float* in, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 r = sample_ambient_light(in + 6 * (size_t)i, in[6 * (size_t)i + 4], in[6 * (size_t)i + 5]);
    out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
}
__global__ void k_hc_scalars(const float* x, int n, const float* tables, float zn, float zf, uint32_t* codes, float* linear) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ColorTables ct{tables};
    codes[i] = srgb_encode(ct, x[i]);
    linear[i] = depth_to_linear(x[i], zn, zf);
}
// Every f32 bit pattern: the device's srgb_encode (search from a hardware log2/exp2 estimate)
// against the definition, the largest code whose threshold is at most x, found by a plain scan.
__global__ void k_hc_srgb_all(const float* tables, unsigned long long* mismatches) {
    ColorTables ct{tables};
    unsigned long long bad = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32);
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = u2f((uint32_t)b);
        uint32_t want = 0;
        if (x >= 1.0f) want = 255;
        else if (x > 0.0f) {
            for (uint32_t c = 1; c < 256 && x >= ct.thresh(c); ++c) want = c;
        }
        if (srgb_encode(ct, x) != want) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
// dst code word + 4 source floats + blend flag in, packed code word out
__global__ void k_hc_blend(const uint32_t* dst, const float* src, const int* blend, int n, const float* tables, uint32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ColorTables ct{tables};
    out[i] = blend_pack(ct, dst[i], V4{src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]}, blend[i] != 0);
}
// hc_setup_triangle per thread: 12 clip floats in, up to 8 sub-triangles of 28 words out
__global__ void k_hc_setup(const float* clip, int n, unsigned W, unsigned H, int shadow, unsigned* out, int* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    counts[i] = setup_triangle_words(clip + 12 * (size_t)i, W, H, shadow, out + (size_t)i * 8 * 28, 8);
}

namespace {
// Device buffers of one call: freed when the call returns, whatever path it takes.
struct DevBufs {
    void* p[8] = {};
    int n = 0;
    hipError_t err = hipSuccess;
    template <typename T> T* make(size_t count, const T* init = nullptr) {
        void* d = nullptr;
        if (err == hipSuccess) err = hipMalloc(&d, count * sizeof(T) + 16);
        if (err == hipSuccess && init) err = hipMemcpy(d, init, count * sizeof(T), hipMemcpyHostToDevice);
        if (d) p[n++] = d;
        return (T*)d;
    }
    template <typename T> void fetch(T* host, const T* dev, size_t count) {
        if (err == hipSuccess) err = hipDeviceSynchronize();
        if (err == hipSuccess) err = hipMemcpy(host, dev, count * sizeof(T), hipMemcpyDeviceToHost);
    }
    ~DevBufs() { for (int i = 0; i < n; ++i) (void)hipFree(p[i]); }
};
}  // namespace

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
// Hash tables built on the host over the world-space box [lo, hi] (every level whose lattice box
// has at most `cap` entries), in `pool`.  Returns the number of tabled levels.
static int build_box_tables(const FrameConsts& fc, const float* lo, const float* hi, int cap, NoiseTables& nt, float* pool,
                            size_t pool_entries, size_t* used_out) {
    memset(&nt, 0, sizeof(nt));
    noise_tables_clear(nt);
    size_t used = 0; int nlev = 0;
    const double drift[3] = {fc.drift_xy, fc.drift_xy, fc.drift_z};
    for (int k = 0; k < 2; ++k) for (int o = 0; o < 8; ++o) {
        double s = k == 0 ? 0.01 : 0.05, m = (double)(1 << o);
        long l[3], n[3];
        for (int r = 0; r < 3; ++r) {
            double a = (lo[r] * s + drift[r]) * m, b = (hi[r] * s + drift[r]) * m;
            l[r] = (long)floor(a) - 1; n[r] = (long)floor(b) + 2 - l[r] + 1;
        }
        size_t entries = (size_t)(n[0] * n[1] * n[2]);
        if ((long)entries > (long)cap || used + entries > pool_entries) continue;
        NoiseTable& t = nt.level[k * 8 + o];
        t.data = pool + used; t.lox = (int)l[0]; t.loy = (int)l[1]; t.loz = (int)l[2];
        t.nx = (int)n[0]; t.ny = (int)n[1]; t.nz = (int)n[2];
        noise_table_seal(t);
        for (size_t i = 0; i < entries; ++i) {
            int ix = (int)(i % n[0]), iy = (int)((i / n[0]) % n[1]), iz = (int)(i / (n[0] * n[1]));
            pool[used + i] = lattice_hash((float)(t.lox + ix), (float)(t.loy + iy), (float)(t.loz + iz));
        }
        used += entries; ++nlev;
    }
    if (used_out) *used_out = used;
    return nlev;
}
static FrameConsts noise_consts(float time) {
    FrameConsts fc; memset(&fc, 0, sizeof(fc));
    fc.time = time; fc.drift_xy = time * 0.2f; fc.drift_z = wr_sin(time * 0.1f) * 0.5f;
    return fc;
}
// fog_noise through hash tables built on the host for a box around p, to compare with the direct
// evaluation (host build of the lookup).
float hc_fog_noise_tab(const float* p, float time, int cap, int* tabled_levels) {
    const FrameConsts fc = noise_consts(time);
    NoiseTables nt;
    static float pool[1 << 16];
    const float lo[3] = {p[0] - 3.0f, p[1] - 3.0f, p[2] - 3.0f}, hi[3] = {p[0] + 3.0f, p[1] + 3.0f, p[2] + 3.0f};
    *tabled_levels = build_box_tables(fc, lo, hi, cap, nt, pool, sizeof(pool) / 4, nullptr);
    return fog_noise_tab(fc, nt, v3(p[0], p[1], p[2]));
}
// The same on the device: tables over the box [lo, hi] built on the host and copied over, then
// fog_noise_tab and fog_noise of n points (inside the box and outside it) by a kernel.
int hc_dev_fog_noise_tab(const float* pts, int n, const float* lo, const float* hi, float time, int cap, float* out_tab,
                         float* out_direct, int* tabled_levels) {
    const FrameConsts fc = noise_consts(time);
    NoiseTables nt;
    const size_t pool_entries = (size_t)1 << 22;
    float* pool = new float[pool_entries];
    size_t used = 0;
    *tabled_levels = build_box_tables(fc, lo, hi, cap, nt, pool, pool_entries, &used);
    DevBufs b;
    float* dpool = b.make<float>(used ? used : 1, pool);
    for (int l = 0; l < 16; ++l)
        if (nt.level[l].data) nt.level[l].data = dpool + (nt.level[l].data - pool);
    delete[] pool;
    float* dp = b.make<float>(3 * (size_t)n, pts);
    float* dt = b.make<float>((size_t)n);
    float* dd = b.make<float>((size_t)n);
    if (b.err == hipSuccess) k_hc_noise_tab<<<(n + 255) / 256, 256>>>(dp, n, fc, nt, dt, dd);
    b.fetch(out_tab, dt, (size_t)n);
    b.fetch(out_direct, dd, (size_t)n);
    return (int)b.err;
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

int hc_setup_triangle(const float* clip, unsigned W, unsigned H, int shadow, unsigned* out, int max_subs) {
    return setup_triangle_words(clip, W, H, shadow, out, max_subs);
}
// Device forms: every function returns 0 or the HIP error code.
int hc_dev_noise(const float* p, int n, float time, float* out) {
    DevBufs b;
    float* dp = b.make<float>(3 * (size_t)n, p);
    float* dout = b.make<float>(3 * (size_t)n);
    if (b.err == hipSuccess) k_hc_noise<<<(n + 255) / 256, 256>>>(dp, n, time, wr_sin(time * 0.1f) * 0.5f, dout);
    b.fetch(out, dout, 3 * (size_t)n);
    return (int)b.err;
}
int hc_dev_brdf(const float* in, int n, float* out) {
    DevBufs b;
    float* din = b.make<float>(17 * (size_t)n, in);
    float* dout = b.make<float>(3 * (size_t)n);
    if (b.err == hipSuccess) k_hc_brdf<<<(n + 255) / 256, 256>>>(din, n, dout);
    b.fetch(out, dout, 3 * (size_t)n);
    return (int)b.err;
}
int hc_dev_ambient(const float* in, int n, float* out) {
    DevBufs b;
    float* din = b.make<float>(6 * (size_t)n, in);
    float* dout = b.make<float>(3 * (size_t)n);
    if (b.err == hipSuccess) k_hc_ambient<<<(n + 255) / 256, 256>>>(din, n, dout);
    b.fetch(out, dout, 3 * (size_t)n);
    return (int)b.err;
}
int hc_dev_scalars(const float* x, int n, float zn, float zf, unsigned* codes, float* linear) {
    tables();
    DevBufs b;
    float* dx = b.make<float>((size_t)n, x);
    float* dt = b.make<float>(768, g_tables);
    uint32_t* dc = b.make<uint32_t>((size_t)n);
    float* dl = b.make<float>((size_t)n);
    if (b.err == hipSuccess) k_hc_scalars<<<(n + 255) / 256, 256>>>(dx, n, dt, zn, zf, dc, dl);
    b.fetch(codes, dc, (size_t)n);
    b.fetch(linear, dl, (size_t)n);
    return (int)b.err;
}
long long hc_dev_srgb_encode_mismatches() {
    tables();
    DevBufs b;
    float* dt = b.make<float>(768, g_tables);
    unsigned long long zero = 0, m = 0;
    unsigned long long* dm = b.make<unsigned long long>(1, &zero);
    if (b.err == hipSuccess) k_hc_srgb_all<<<8192, 256>>>(dt, dm);
    b.fetch(&m, dm, 1);
    return b.err == hipSuccess ? (long long)m : -1;
}
int hc_dev_blend_pack(const unsigned* dst, const float* src, const int* blend, int n, unsigned* out) {
    tables();
    DevBufs b;
    uint32_t* dd = b.make<uint32_t>((size_t)n, dst);
    float* ds = b.make<float>(4 * (size_t)n, src);
    int* db = b.make<int>((size_t)n, blend);
    float* dt = b.make<float>(768, g_tables);
    uint32_t* dout = b.make<uint32_t>((size_t)n);
    if (b.err == hipSuccess) k_hc_blend<<<(n + 255) / 256, 256>>>(dd, ds, db, n, dt, dout);
    b.fetch(out, dout, (size_t)n);
    return (int)b.err;
}
// n triangles of 12 clip floats; out holds 8 * 28 words per triangle, counts one int per triangle
int hc_dev_setup_triangles(const float* clip, int n, unsigned W, unsigned H, int shadow, unsigned* out, int* counts) {
    DevBufs b;
    float* dc = b.make<float>(12 * (size_t)n, clip);
    unsigned* dout = b.make<unsigned>((size_t)n * 8 * 28);
    int* dn = b.make<int>((size_t)n);
    if (b.err == hipSuccess) b.err = hipMemset(dout, 0, (size_t)n * 8 * 28 * 4);
    if (b.err == hipSuccess) k_hc_setup<<<(n + 255) / 256, 256>>>(dc, n, W, H, shadow, dout, dn);
    b.fetch(out, dout, (size_t)n * 8 * 28);
    b.fetch(counts, dn, (size_t)n);
    return (int)b.err;
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
