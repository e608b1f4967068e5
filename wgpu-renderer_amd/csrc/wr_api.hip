// wr_api.hip -- C ABI of libwr_hip.so (declared in include/wr_abi.h): context, scene
// upload, per-frame kernel sequence, read-back, and the host scene math that mirrors
// src/core/camera.rs, light.rs and instance.rs.
//
// Device memory layout (all hipMalloc, owned by the context):
//   scene   : per model SoA vertex streams (pos float4, uv float2, normal/tangent/bitangent
//             float4), one merged u32 index buffer with mesh bases applied, one material id
//             per triangle; all textures in one RGBA8 texel pool with a descriptor table.
//   frame   : geometry vertex-stage outputs (5 float4 streams), shadow clip positions
//             [6][V], triangle records + bin lists per raster target, visibility (u32) and
//             depth (f32) buffers, RGBA8 colour, shadow array [6][S][S] f32.
// There is no CPU rendering path: every entry point that produces pixels launches the
// kernels of wr_kernels.h and fails with WR_ERR_DEVICE if HIP is unusable.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wr_abi.h"
#include "wr_kernels.h"

using namespace wr;

struct ncclUniqueIdBytes { char internal[128]; };  // ncclUniqueId of rccl.h, passed by value

static_assert(sizeof(WrModelVertex) == 56, "ModelVertex layout (model.rs:14-22)");
static_assert(sizeof(WrInstanceRaw) == 100, "InstanceRaw layout (instance.rs:9-14)");
static_assert(sizeof(WrCameraUniform) == 224, "CameraUniform layout (camera.rs:104-112)");
static_assert(sizeof(WrLightUniform) == 416, "LightUniform layout (light.rs:11-18)");
static_assert(sizeof(WrGlobalUniforms) == 16, "GlobalUniforms layout (state.rs:21-28)");
static_assert(sizeof(WrMaterialUniform) == 16, "MaterialUniform layout (material.rs:13-18)");
#ifndef WR_COLOR_RASTER_THREADS
#define WR_COLOR_RASTER_THREADS 256
#endif
constexpr int kColorRasterThreads = WR_COLOR_RASTER_THREADS;
// With one frame in flight the geometry raster is on the frame's critical path and its longest
// bins (dozens of bin-filling triangles, one per wavefront at a time) set its length: eight
// wavefronts per bin instead of four.  With frames overlapping the smaller workgroup is better.
constexpr int kColorRasterThreadsAlone = 512;
constexpr int kLayerRasterThreadsAlone = 1024;
constexpr uint32_t kProfEveryAlone = 4;  // wr_set_profiling level 2, one frame in flight: frames per recorded event pair
#ifndef WR_SHADOW_RASTER_THREADS
#define WR_SHADOW_RASTER_THREADS 256
#endif
constexpr int kShadowRasterThreads = WR_SHADOW_RASTER_THREADS;
#ifndef WR_NOISE_LEVEL_ENTRIES
#define WR_NOISE_LEVEL_ENTRIES (4u << 20)
#endif
#ifndef WR_NOISE_POOL_ENTRIES
#define WR_NOISE_POOL_ENTRIES (12u << 20)
#endif
constexpr size_t kNoiseLevelEntries = WR_NOISE_LEVEL_ENTRIES;  // largest hash table of one noise level
constexpr size_t kNoisePoolEntries = WR_NOISE_POOL_ENTRIES;    // all levels together (4 bytes each, per frame in flight)
static_assert(kNoiseLevelEntries <= kNoiseTableMaxEntries, "value_noise_tab forms a table index in f32");
static_assert(sizeof(FrameConsts) < 3072, "FrameConsts must fit the kernel argument segment");

// Work buffers of wr_render_frame_static, the entry point for callers that cannot allocate.
// One copy per device that loads the library, held until the process ends.  The name is a
// host-side stand-in: host code reaches the memory only through g_scratch_ptr[device], taken
// once per device with hipGetSymbolAddress in wr_create.
constexpr size_t kStaticScratchBytes = (size_t)96 << 20;
constexpr size_t kStaticNoisePoolEntries = (size_t)1 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxScratchDevices = 64;
static unsigned char* g_scratch_ptr[kMaxScratchDevices];
// Every static frame on a device uses that device's one array, from whatever context and host
// thread it comes: g_static_done[device] is recorded behind each of them and the next one waits
// for it on the device before it touches the array (submission itself is under the mutex).
static hipEvent_t g_static_done[kMaxScratchDevices];
static bool g_static_recorded[kMaxScratchDevices];
static std::mutex g_scratch_mutex;

namespace {

struct ModelHost {
    std::vector<WrModelVertex> verts;
    std::vector<uint32_t> idx;
    std::vector<uint32_t> tri_mat;
    std::vector<MatDesc> mats;
    WrInstanceRaw inst;
    bool has_inst = false;
};
struct ModelDev {
    float4* pos = nullptr;
    float2* uv = nullptr;
    float4 *nrm = nullptr, *tan = nullptr, *bit = nullptr;
    uint32_t* idx = nullptr;
    uint32_t* tri_mat = nullptr;
    uint32_t nv = 0, ntris = 0;
};
struct TexHost { uint32_t w, h; int srgb; std::vector<uint8_t> px; };

// Where a frame's work buffers come from: base == nullptr makes every block its own
// hipMalloc (contexts created by wr_create); otherwise blocks are carved, 256-byte aligned,
// from [base, base + cap) and nothing is allocated (wr_render_frame_static).  With
// dry == true nothing is handed out, only `off` advances: the size a set of buffers needs.
struct Arena {
    uint8_t* base = nullptr;
    size_t off = 0, cap = 0;
    bool dry = false;
    bool carving() const { return base != nullptr || dry; }
};
template <class T>
hipError_t arena_take(Arena& a, T*& p, size_t bytes) {
    if (!a.carving()) return hipMalloc(&p, bytes);
    const size_t o = (a.off + 255) & ~(size_t)255;
    a.off = o + bytes;
    if (a.dry) { p = nullptr; return hipSuccess; }
    if (a.off > a.cap) return hipErrorOutOfMemory;
    p = reinterpret_cast<T*>(a.base + o);
    return hipSuccess;
}

struct WsAlloc {
    RasterWS ws{};
    uint32_t* counters = nullptr;  // ovf_count | bin_count | bin_cursor, cleared every frame
    size_t counters_bytes = 0;
};

}  // namespace

// Everything one frame in flight writes.  With several of these, consecutive frames rotate
// through the sets (each with its own pair of streams), so the early, latency-bound stages of
// one frame run under the fog march of the frames before it.
struct FrameRes {
    // ms: geometry pass, shading, fog blend.  side: shadow pass.  fs: fog volume set-up and noise
    // tables, then (once the geometry depth exists) the marches, next to the shading kernels.
    hipStream_t ms = nullptr, side = nullptr, fs = nullptr;
    hipEvent_t ev_fork = nullptr, ev_start = nullptr, ev_join = nullptr, ev_shadow_done = nullptr, ev_geom_done = nullptr, ev_done = nullptr;
    hipEvent_t ev_prologue_done = nullptr;  // fog volume set-up and noise tables done (one frame in flight)
    hipEvent_t ev_fog_volume_done = nullptr;  // fog volume set-up done (one frame in flight)
    hipEvent_t ev_shade_tail = nullptr;       // the frame's shading is down to its last kernel (one frame in flight)
    hipEvent_t ev_targets_free = nullptr;     // the frame has read its depth and shadow maps for the last time (one frame in flight)
    hipStream_t cur_ms = nullptr;           // the stream the frame in these buffers ends on (see frame_stages)
    bool submitted = false;  // ev_done has been recorded
    uint64_t serial = 0;     // number (from 1) of the last frame submitted to this set
    // d_shadow holds the cleared depth (1.0 everywhere), the value a frame without shadow passes
    // samples: as at allocation, until a shadow pass draws into it
    bool shadow_is_clear = true;

    VertexOut vout{};
    WsAlloc geom, shadow, fog;
    uint32_t* d_shadow_pairs = nullptr;  // [6 * triangles] (triangle, face) pairs the shadow passes set up this frame
    uint32_t* d_vis = nullptr;
    float* d_depth = nullptr;
    uint32_t* d_color = nullptr;
    float* d_shadow = nullptr;
    BlendWS blend{};                 // ordered-blend work space, sized with the colour target
    size_t blend_clear_bytes = 0;    // counts + bin flags, cleared every frame
    float* d_noise_pool = nullptr;   // hash tables of the low noise octaves, rebuilt every frame
    size_t noise_pool_entries = 0;   // capacity of d_noise_pool
    uint32_t* d_fog_pixel = nullptr; // [W*H] fog march work items: pixel index
    float4* d_fog_ray = nullptr;     // [W*H] ray origin + max_fog_depth
    float4* d_fog_res = nullptr;     // [W*H] view march result of each item
    float* d_fog_noise_view = nullptr;   // [kFogMaxSteps][W*H] fog_noise of the view-march steps
    float* d_fog_noise_light = nullptr;  // [kFogLightMaxSteps][W*H] fog_noise of the light-march steps
    uint32_t* d_fog_lit = nullptr;   // [W*H] items whose fog point is lit
    uint32_t* d_fog_multi = nullptr; // [W*H] pixels with several fog fragments
};

struct WrContext {
    WrConfig cfg{};
    int W = 0, H = 0, S = 0;
    hipStream_t stream = nullptr;  // the caller's stream (or a library-owned one): submission order, read-back
    bool own_stream = false;
    std::string err;

    ModelHost host[3];
    std::vector<TexHost> tex_host;
    bool scene_dirty = true;
    bool broken = false;  // a resize ran out of memory: the targets are gone, nothing can be rendered

    ModelDev dev[3];
    MatDesc* d_mats = nullptr;
    TexDesc* d_tex = nullptr;
    uint32_t* d_texels = nullptr;
    float* d_tables = nullptr;
    float fog_aabb[6] = {0, 0, 0, 0, 0, 0};  // fog model bounds in model space (min xyz, max xyz)

    static constexpr int kMaxFrames = 4;
    FrameRes fr[kMaxFrames];
    int nfr = 1;   // frames in flight (WrConfig.frames_in_flight)
    int cur = 0;   // the set the most recent frame was submitted to
    uint32_t* d_gather_send = nullptr;  // bin-sharded contexts: this rank's packed bins
    uint32_t* d_gather_recv = nullptr;  // and, for the gathering rank, every rank's slot
    unsigned long long* d_counters = nullptr;
    // Per set of frame buffers four words: [0] error bits, [1] geometry bin items, [2] shadow bin
    // items, [3] fog (unused).  A set's words belong to the frames rendered on that set, so an
    // overflow names its frame.  One more group of four, after the sets', is wr_render_frame_static's.
    uint32_t* d_err = nullptr;
    uint64_t frames_submitted = 0;
    uint64_t static_frames_submitted = 0;
    // The library has queued work of its own on `stream` (counter reset, bin pack / gather, the
    // static frame's clears) since the last frame forked from it.
    bool stream_has_work = true;
    bool flip_streams = false;  // one frame in flight: the main and the fog stream change roles with every frame

    int ev_level = 0;   // level the recorded frames' events were taken at
    int profiling = 0;  // 0 off, 1 every stage's events, 2 only the events around the fog noise / march kernels
    bool counting = false;
    // Profiling keeps one set of stage-boundary events per frame in a ring; wr_get_stats
    // averages the stage times over the frames recorded since profiling was switched on.
    static constexpr int kEvFrames = 128;
    static constexpr int kEvPerFrame = 11;  // see frame_stages
    hipEvent_t ev[kEvFrames][kEvPerFrame] = {};
    int ev_count = 0;  // frames recorded since wr_set_profiling(1), saturates at kEvFrames
    uint32_t prof_tick = 0;  // frames submitted since wr_set_profiling (level 2 samples them with one frame in flight)
    int ev_next = 0;
    WrFrameStats stats{};
};

namespace {

int fail(WrContext* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define WR_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? WR_ERR_OUT_OF_MEMORY : WR_ERR_DEVICE, \
                        std::string(#call) + ": " + hipGetErrorString(e_));                  \
    } while (0)

template <class T>
void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_ws(WsAlloc& w) {
    dfree(w.ws.recs); dfree(w.ws.bb); dfree(w.ws.betas); dfree(w.counters); dfree(w.ws.bin_offset); dfree(w.ws.bin_items);
    dfree(w.ws.clip_queue);
    w = WsAlloc();
}

int alloc_ws(WrContext* ctx, Arena& ar, WsAlloc& w, uint32_t slots, uint32_t ovf_cap, uint32_t items_cap, int bins_x, int bins_y,
             int layers, uint32_t* err, uint32_t* total_items, size_t extra_counters = 0) {
    if (!ar.carving()) free_ws(w);
    RasterWS& ws = w.ws;
    ws.slots = slots; ws.ovf_cap = ovf_cap; ws.items_cap = items_cap;
    ws.bins_x = bins_x; ws.bins_y = bins_y; ws.layers = layers;
    ws.err = err; ws.total_items = total_items;
    size_t nb = (size_t)bins_x * bins_y * layers;
    WR_HIP(arena_take(ar, ws.recs, sizeof(Rec) * (size_t)layers * ((size_t)slots + ovf_cap)));
    WR_HIP(arena_take(ar, ws.bb, sizeof(uint2) * (size_t)layers * ((size_t)slots + ovf_cap)));
    WR_HIP(arena_take(ar, ws.betas, sizeof(float) * 9 * (size_t)layers * ovf_cap + 64));
    w.counters_bytes = sizeof(uint32_t) * (2 * (size_t)layers + 2 * nb + extra_counters + 1);
    WR_HIP(arena_take(ar, w.counters, w.counters_bytes));
    if (ar.dry) {  // sizing pass: no pointers to derive
        size_t dummy_bytes[3] = {sizeof(uint32_t) * (size_t)layers * slots, sizeof(uint32_t) * (nb + 1), sizeof(uint32_t) * ((size_t)items_cap + 1)};
        void* dummy = nullptr;
        for (size_t b : dummy_bytes) WR_HIP(arena_take(ar, dummy, b));
        return WR_OK;
    }
    ws.ovf_count = w.counters;
    ws.bin_count = w.counters + layers;
    ws.bin_cursor = ws.bin_count + nb;
    ws.clip_count = ws.bin_cursor + nb;
    ws.scan_total = w.counters + 2 * (size_t)layers + 2 * nb + extra_counters;
    WR_HIP(arena_take(ar, ws.clip_queue, sizeof(uint32_t) * (size_t)layers * slots));
    WR_HIP(arena_take(ar, ws.bin_offset, sizeof(uint32_t) * (nb + 1)));
    WR_HIP(arena_take(ar, ws.bin_items, sizeof(uint32_t) * ((size_t)items_cap + 1)));
    // (the counters are cleared at the start of every frame that uses them)
    if (!ar.carving()) WR_HIP(hipMemset(w.counters, 0, w.counters_bytes));
    return WR_OK;
}

void free_targets(WrContext* c) {
    for (FrameRes& f : c->fr) {
        dfree(f.d_vis); dfree(f.d_depth); dfree(f.d_color); dfree(f.d_fog_pixel); dfree(f.d_fog_ray); dfree(f.d_fog_res); dfree(f.d_fog_noise_view); dfree(f.d_fog_noise_light); dfree(f.d_fog_lit); dfree(f.d_fog_multi);
        BlendWS& b = f.blend;
        dfree(b.rp_slot); dfree(b.slot_pixel); dfree(b.counts); dfree(b.lists); dfree(b.list_count);
        dfree(b.slow_slots);
        b = BlendWS();
    }
}
void free_scene(WrContext* c) {
    for (int m = 0; m < 3; ++m) {
        ModelDev& d = c->dev[m];
        dfree(d.pos); dfree(d.uv); dfree(d.nrm); dfree(d.tan); dfree(d.bit); dfree(d.idx); dfree(d.tri_mat);
        d = ModelDev();
    }
    dfree(c->d_mats); dfree(c->d_tex); dfree(c->d_texels);
    for (FrameRes& f : c->fr) {
        dfree(f.vout.clip); dfree(f.vout.v0); dfree(f.vout.v1); dfree(f.vout.v2); dfree(f.vout.v3);
        free_ws(f.geom); free_ws(f.shadow); free_ws(f.fog);
        dfree(f.d_shadow_pairs);
    }
}

// The buffers of one frame in flight that depend on the target size.  `color` == false leaves
// d_color alone (the caller of wr_render_frame_static brings the colour target).
int alloc_frame_targets(WrContext* ctx, Arena& ar, FrameRes* f, bool color) {
    const size_t n = (size_t)ctx->W * ctx->H;
    WR_HIP(arena_take(ar, f->d_vis, n * 4));
    WR_HIP(arena_take(ar, f->d_depth, n * 4));
    if (color) WR_HIP(arena_take(ar, f->d_color, n * 4));
    // Ordered blending: every pixel can be flagged, and every flagged pixel keeps up to
    // kReplayK covering fragments (128 bytes per pixel of the target; only flagged pixels'
    // lists are ever touched).
    BlendWS& b = f->blend;
    const size_t nbins = (size_t)((ctx->W + kBin - 1) / kBin) * (size_t)((ctx->H + kBin - 1) / kBin);
    b.cap = (uint32_t)n;
    f->blend_clear_bytes = sizeof(uint32_t) * (4 + nbins);
    WR_HIP(arena_take(ar, b.rp_slot, n * 4));
    WR_HIP(arena_take(ar, b.slot_pixel, n * 4));
    WR_HIP(arena_take(ar, b.counts, f->blend_clear_bytes));
    b.bin_flag = b.counts ? b.counts + 4 : nullptr;
    WR_HIP(arena_take(ar, b.lists, n * kReplayK * sizeof(uint2)));
    WR_HIP(arena_take(ar, b.list_count, n * 4));
    WR_HIP(arena_take(ar, b.slow_slots, n * 4));
    WR_HIP(arena_take(ar, f->d_fog_pixel, n * 4));
    WR_HIP(arena_take(ar, f->d_fog_ray, n * sizeof(float4)));
    WR_HIP(arena_take(ar, f->d_fog_res, n * sizeof(float4)));
    WR_HIP(arena_take(ar, f->d_fog_noise_view, n * sizeof(float) * kFogMaxSteps));
    WR_HIP(arena_take(ar, f->d_fog_noise_light, n * sizeof(float) * kFogLightMaxSteps));
    WR_HIP(arena_take(ar, f->d_fog_lit, n * 4));
    WR_HIP(arena_take(ar, f->d_fog_multi, n * 4));
    return WR_OK;
}

int alloc_targets(WrContext* ctx) {
    free_targets(ctx);
    const size_t n = (size_t)ctx->W * ctx->H;
    dfree(ctx->d_gather_send); dfree(ctx->d_gather_recv);
    if (ctx->cfg.tile_world > 1) {
        uint32_t owned = 0, slot = 0;
        wr_gather_layout((uint32_t)ctx->W, (uint32_t)ctx->H, ctx->cfg.tile_world, ctx->cfg.tile_rank, &owned, &slot);
        const size_t bin_bytes = (size_t)kBin * kBin * 4;
        WR_HIP(hipMalloc(&ctx->d_gather_send, std::max<size_t>(slot, 1) * bin_bytes));
        WR_HIP(hipMalloc(&ctx->d_gather_recv, std::max<size_t>((size_t)slot * ctx->cfg.tile_world, 1) * bin_bytes));
    }
    for (int k = 0; k < ctx->nfr; ++k) {
        FrameRes* f = &ctx->fr[k];
        Arena ar;
        int rc = alloc_frame_targets(ctx, ar, f, true);
        if (rc != WR_OK) return rc;
        WR_HIP(hipMemset(f->blend.rp_slot, 0, n * 4));
        WR_HIP(hipMemset(f->d_color, 0, n * 4));
        // bins a sharded context never renders keep defined contents: nothing drawn, far plane
        WR_HIP(hipMemset(f->d_vis, 0xFF, n * 4));
        hipLaunchKernelGGL(k_fill_f32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, f->d_depth, n, 1.0f);
        WR_HIP(hipStreamSynchronize(ctx->stream));
    }
    return WR_OK;
}

// Waits for every frame in flight and for the caller's stream.
int sync_all(WrContext* ctx) {
    for (int k = 0; k < ctx->nfr; ++k) {
        if (ctx->fr[k].ms) WR_HIP(hipStreamSynchronize(ctx->fr[k].ms));
        if (ctx->fr[k].side) WR_HIP(hipStreamSynchronize(ctx->fr[k].side));
        if (ctx->fr[k].fs) WR_HIP(hipStreamSynchronize(ctx->fr[k].fs));
    }
    WR_HIP(hipStreamSynchronize(ctx->stream));
    return WR_OK;
}

// The buffers of one frame in flight that depend on the uploaded scene: vertex-stage outputs
// and the three raster work spaces.
int alloc_frame_scene(WrContext* ctx, Arena& ar, FrameRes* f) {
    const uint32_t nvg = ctx->dev[WR_MODEL_GEOMETRY].nv, ntg = ctx->dev[WR_MODEL_GEOMETRY].ntris;
    const uint32_t ntl = ctx->dev[WR_MODEL_LIGHT_MARKER].ntris, ntf = ctx->dev[WR_MODEL_FOG_VOLUME].ntris;
    const size_t nvs = std::max<uint32_t>(nvg, 1);
    // (a frame carved from the static buffer is not one of the context's sets: it has error words of its own)
    const ptrdiff_t set = (f >= &ctx->fr[0] && f < &ctx->fr[WrContext::kMaxFrames]) ? f - &ctx->fr[0] : WrContext::kMaxFrames;
    uint32_t* const err = ctx->d_err + 4 * set;
    WR_HIP(arena_take(ar, f->vout.clip, sizeof(float4) * nvs));
    WR_HIP(arena_take(ar, f->vout.v0, sizeof(float4) * nvs));
    WR_HIP(arena_take(ar, f->vout.v1, sizeof(float4) * nvs));
    WR_HIP(arena_take(ar, f->vout.v2, sizeof(float4) * nvs));
    WR_HIP(arena_take(ar, f->vout.v3, sizeof(float4) * nvs));
    int rc;
    // capacities in 64 bits: the sums pass 2^32 for meshes well below the 2^29-triangle limit
    {
        const uint64_t slots = (uint64_t)ntg + ntl;
        const uint64_t ovf = slots / 2 + 8192;
        const int bx = (ctx->W + kBin - 1) / kBin, by = (ctx->H + kBin - 1) / kBin;
        uint64_t items = 8u * (slots + ovf) + 64u * (uint64_t)(bx * by) + (1u << 20);
        // Test aid: WR_DEBUG_BIN_ITEMS_CAP shrinks the geometry pass's bin item list, so that the
        // overflow report (wr_sync) can be exercised; every write to the list is bounds-checked.
        if (const char* cap_env = getenv("WR_DEBUG_BIN_ITEMS_CAP")) items = std::min<uint64_t>(items, (uint64_t)std::max(atoll(cap_env), 1ll));
        if (slots + ovf > 0xFFFFFFF0ull || items > 0xFFFFFFF0ull) return fail(ctx, WR_ERR_UNSUPPORTED, "geometry too large for the 32-bit bin lists");
        if ((rc = alloc_ws(ctx, ar, f->geom, (uint32_t)std::max<uint64_t>(slots, 1), (uint32_t)ovf, (uint32_t)items, bx, by, 1, err, err + 1)) != WR_OK) return rc;
    }
    {
        const uint64_t slots = std::max(ntg, 1u);
        const uint64_t ovf = slots / 4 + 8192;
        const int bs = (ctx->S + kBin - 1) / kBin;
        const uint64_t items = 6u * (4u * slots + 64u * (uint64_t)(bs * bs) + (1u << 18));
        if (6 * (slots + ovf) > 0xFFFFFFF0ull || items > 0xFFFFFFF0ull) return fail(ctx, WR_ERR_UNSUPPORTED, "geometry too large for the 32-bit bin lists");
        // (one extra counter word behind the per-face clip counts: the length of the pair list)
        if ((rc = alloc_ws(ctx, ar, f->shadow, (uint32_t)slots, (uint32_t)ovf, (uint32_t)items, bs, bs, 6, err, err + 2, 1)) != WR_OK) return rc;
        if (!ar.carving()) dfree(f->d_shadow_pairs);
        WR_HIP(arena_take(ar, f->d_shadow_pairs, sizeof(uint32_t) * 6 * (size_t)slots));
    }
    {
        const uint32_t slots = std::max(ntf, 1u);
        if ((rc = alloc_ws(ctx, ar, f->fog, slots, slots * kMaxFan + 8, 8, 1, 1, 1, err, err + 3, 4)) != WR_OK) return rc;
    }
    return WR_OK;
}

// (Re)builds every device buffer that depends on the uploaded scene or the target size.
int build_scene(WrContext* ctx) {
    WR_HIP(hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != WR_OK) return rcs;
    }
    free_scene(ctx);
    for (int m = 0; m < 3; ++m) {
        ModelHost& h = ctx->host[m];
        ModelDev& d = ctx->dev[m];
        d.nv = (uint32_t)h.verts.size();
        d.ntris = (uint32_t)(h.idx.size() / 3);
        if (d.nv == 0 || d.ntris == 0) continue;
        std::vector<float4> pos(d.nv), nrm(d.nv), tan(d.nv), bit(d.nv);
        std::vector<float2> uv(d.nv);
        for (uint32_t i = 0; i < d.nv; ++i) {
            const WrModelVertex& v = h.verts[i];
            pos[i] = make_float4(v.position[0], v.position[1], v.position[2], 1.0f);
            uv[i] = make_float2(v.tex_coords[0], v.tex_coords[1]);
            nrm[i] = make_float4(v.normal[0], v.normal[1], v.normal[2], 0.0f);
            tan[i] = make_float4(v.tangent[0], v.tangent[1], v.tangent[2], 0.0f);
            bit[i] = make_float4(v.bitangent[0], v.bitangent[1], v.bitangent[2], 0.0f);
        }
        WR_HIP(hipMalloc(&d.pos, sizeof(float4) * d.nv));
        WR_HIP(hipMalloc(&d.uv, sizeof(float2) * d.nv));
        WR_HIP(hipMalloc(&d.nrm, sizeof(float4) * d.nv));
        WR_HIP(hipMalloc(&d.tan, sizeof(float4) * d.nv));
        WR_HIP(hipMalloc(&d.bit, sizeof(float4) * d.nv));
        WR_HIP(hipMalloc(&d.idx, sizeof(uint32_t) * h.idx.size()));
        WR_HIP(hipMalloc(&d.tri_mat, sizeof(uint32_t) * d.ntris));
        WR_HIP(hipMemcpy(d.pos, pos.data(), sizeof(float4) * d.nv, hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.uv, uv.data(), sizeof(float2) * d.nv, hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.nrm, nrm.data(), sizeof(float4) * d.nv, hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.tan, tan.data(), sizeof(float4) * d.nv, hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.bit, bit.data(), sizeof(float4) * d.nv, hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.idx, h.idx.data(), sizeof(uint32_t) * h.idx.size(), hipMemcpyHostToDevice));
        WR_HIP(hipMemcpy(d.tri_mat, h.tri_mat.data(), sizeof(uint32_t) * d.ntris, hipMemcpyHostToDevice));
    }
    {
        const ModelHost& fh = ctx->host[WR_MODEL_FOG_VOLUME];
        for (int k = 0; k < 3; ++k) { ctx->fog_aabb[k] = 0.0f; ctx->fog_aabb[3 + k] = 0.0f; }
        for (size_t i = 0; i < fh.verts.size(); ++i)
            for (int k = 0; k < 3; ++k) {
                float v = fh.verts[i].position[k];
                if (i == 0 || v < ctx->fog_aabb[k]) ctx->fog_aabb[k] = v;
                if (i == 0 || v > ctx->fog_aabb[3 + k]) ctx->fog_aabb[3 + k] = v;
            }
    }
    // textures: one texel pool
    {
        std::vector<TexDesc> descs;
        size_t total = 0;
        for (const TexHost& t : ctx->tex_host) {
            descs.push_back(TexDesc{(uint32_t)total, t.w, t.h, (uint32_t)(t.srgb != 0)});
            total += (size_t)t.w * t.h;
        }
        if (total >= 0xFFFFFFFFull) return fail(ctx, WR_ERR_UNSUPPORTED, "texture pool exceeds 2^32 texels");
        WR_HIP(hipMalloc(&ctx->d_texels, std::max<size_t>(total, 1) * 4));
        WR_HIP(hipMalloc(&ctx->d_tex, std::max<size_t>(descs.size(), 1) * sizeof(TexDesc)));
        for (size_t i = 0; i < ctx->tex_host.size(); ++i)
            WR_HIP(hipMemcpy(ctx->d_texels + descs[i].offset, ctx->tex_host[i].px.data(), ctx->tex_host[i].px.size(), hipMemcpyHostToDevice));
        if (!descs.empty()) WR_HIP(hipMemcpy(ctx->d_tex, descs.data(), descs.size() * sizeof(TexDesc), hipMemcpyHostToDevice));
    }
    {
        // a material whose base-colour texture has alpha 255 everywhere can only produce alpha 1
        std::vector<MatDesc>& mats = ctx->host[WR_MODEL_GEOMETRY].mats;
        std::vector<int> tex_opaque(ctx->tex_host.size(), -1);
        for (MatDesc& m : mats) {
            if (tex_opaque[m.base] < 0) {
                const TexHost& t = ctx->tex_host[m.base];
                int op = 1;
                for (size_t i = 3; i < t.px.size(); i += 4) if (t.px[i] != 255) { op = 0; break; }
                tex_opaque[m.base] = op;
            }
            m.opaque = (uint32_t)tex_opaque[m.base];
        }
        WR_HIP(hipMalloc(&ctx->d_mats, std::max<size_t>(mats.size(), 1) * sizeof(MatDesc)));
        if (!mats.empty()) WR_HIP(hipMemcpy(ctx->d_mats, mats.data(), mats.size() * sizeof(MatDesc), hipMemcpyHostToDevice));
    }
    const uint32_t ntg = ctx->dev[WR_MODEL_GEOMETRY].ntris, ntl = ctx->dev[WR_MODEL_LIGHT_MARKER].ntris;
    if ((uint64_t)ntg + ntl >= (1u << 29)) return fail(ctx, WR_ERR_UNSUPPORTED, "more than 2^29 triangles");
    for (int k = 0; k < ctx->nfr; ++k) {
        Arena ar;
        int rc = alloc_frame_scene(ctx, ar, &ctx->fr[k]);
        if (rc != WR_OK) return rc;
    }
    ctx->scene_dirty = false;
    return WR_OK;
}

void build_tables(float* t) {
    auto to_linear = [](double s) { return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4); };
    for (int c = 0; c < 256; ++c) {
        t[c] = (float)to_linear((double)c / 255.0);
        t[256 + c] = (float)c / 255.0f;
        t[512 + c] = c == 0 ? 0.0f : (float)to_linear(((double)c - 0.5) / 255.0);
    }
}

// fog.wgsl:131 samples the geometry depth with a Linear sampler at the fragment's own texel
// centre; in f32, (px + 0.5) / width * W - 0.5 is not always exactly px, and then the
// neighbouring texel enters the blend with a weight of a few 1e-6 (fog_max_depth restates
// that arithmetic).  Returns whether, for any column or row, such a tap lies in another 64-pixel
// bin than the fragment: a bin-sharded context then needs the depth of bins it does not own.
// (1920x1080 and 3840x2160 have no such tap; 1000 or 720 pixels do.)
bool depth_taps_cross_bins(int W, int H, float planes_w, float planes_h) {
    auto axis = [](int n, float extent) {
        for (int p = 0; p < n; ++p) {
            const float u = ((float)p + 0.5f) / extent;
            const float x = guard_coord(u * (float)n - 0.5f);
            const float x0 = floorf(x);
            const int i0 = clampi((int)x0, n), i1 = clampi((int)x0 + 1, n);
            if (i0 / kBin != p / kBin) return true;
            if (x - x0 != 0.0f && i1 / kBin != p / kBin) return true;
        }
        return false;
    };
    return axis(W, planes_w) || axis(H, planes_h);
}

FrameConsts make_consts(const WrContext* ctx, const WrCameraUniform* cam, const WrLightUniform* light,
                        const WrGlobalUniforms* glob) {
    FrameConsts fc;
    memset(&fc, 0, sizeof(fc));
    mat4_mul_mat4(&cam->proj[0][0], &cam->view[0][0], fc.pv);
    memcpy(fc.model_geom, ctx->host[WR_MODEL_GEOMETRY].inst.model, 64);
    memcpy(fc.normal_geom, ctx->host[WR_MODEL_GEOMETRY].inst.normal, 36);
    memcpy(fc.model_fog, ctx->host[WR_MODEL_FOG_VOLUME].inst.model, 64);
    memcpy(fc.light_m, light->matrices, sizeof(fc.light_m));
    for (int k = 0; k < 3; ++k) { fc.cam_pos[k] = cam->position[k]; fc.light_pos[k] = light->position[k]; }
    for (int k = 0; k < 4; ++k) { fc.light_color[k] = light->color[k]; fc.planes[k] = cam->planes[k]; }
    fc.time = glob->time;
    fc.drift_xy = glob->time * 0.2f;
    fc.drift_z = wr_sin(glob->time * 0.1f) * 0.5f;
    fc.use_shadowmaps = glob->use_shadowmaps;
    fc.light_tame = 1u;
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 16; ++k)
            if (!(std::fabs(fc.light_m[i][k]) < 1.0e15f)) fc.light_tame = 0u;
    fc.W = ctx->W; fc.H = ctx->H; fc.S = ctx->S;
    fc.tile_rank = (int)ctx->cfg.tile_rank;
    fc.tile_world = (int)(ctx->cfg.tile_world ? ctx->cfg.tile_world : 1);
    fc.geom_tile_rank = fc.tile_rank; fc.geom_tile_world = fc.tile_world;
    if (fc.tile_world > 1 && depth_taps_cross_bins(ctx->W, ctx->H, cam->planes[2], cam->planes[3])) {
        fc.geom_tile_rank = 0; fc.geom_tile_world = 1;
    }
    return fc;
}

dim3 grid1(uint32_t n, uint32_t layers = 1) { return dim3((n + 255u) / 256u, layers, 1); }

int bin_and_raster(WrContext* ctx, FrameRes* f, hipStream_t st, WsAlloc& w, bool shadow, int tr, int tw,
                   hipEvent_t prev_done = nullptr) {
    RasterWS& ws = w.ws;
    uint32_t nrec = ws.slots + ws.ovf_cap;
    const uint32_t nbins = (uint32_t)(ws.bins_x * ws.bins_y * ws.layers);
    if (ctx->nfr == 1) hipLaunchKernelGGL(k_scan, dim3((nbins + kScanThreads - 1) / kScanThreads), dim3(kScanThreads), 0, st, ws);
    else hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(kScanThreads), 0, st, ws);
    hipLaunchKernelGGL(k_bin_fill, grid1(nrec, ws.layers), dim3(256), 0, st, ws, tr, tw);
    dim3 g((uint32_t)(ws.bins_x * ws.bins_y), (uint32_t)ws.layers, 1);
    if (prev_done) WR_HIP(hipStreamWaitEvent(st, prev_done, 0));  // the previous frame's fog tail reads the target this raster writes
    if (shadow)
        hipLaunchKernelGGL((k_raster<kRasterShadow, kShadowRasterThreads>), g, dim3(kShadowRasterThreads), 0, st, ws, ctx->S, ctx->S, (uint32_t*)nullptr,
                           f->d_shadow, 0, 1, LayerSink{});
    else if (ctx->nfr == 1)
        hipLaunchKernelGGL((k_raster<kRasterColor, kColorRasterThreadsAlone>), g, dim3(kColorRasterThreadsAlone), 0, st, ws, ctx->W, ctx->H, f->d_vis,
                           f->d_depth, tr, tw, LayerSink{});
    else
        hipLaunchKernelGGL((k_raster<kRasterColor, kColorRasterThreads>), g, dim3(kColorRasterThreads), 0, st, ws, ctx->W, ctx->H, f->d_vis,
                           f->d_depth, tr, tw, LayerSink{});
    return WR_OK;
}

int shadow_stage(WrContext* ctx, FrameRes* f, hipStream_t st, const FrameConsts& fc, hipEvent_t prev_done = nullptr) {
    const ModelDev& g = ctx->dev[WR_MODEL_GEOMETRY];
    ShadowXf xf;
    memcpy(xf.model, fc.model_geom, sizeof(xf.model));
    memcpy(xf.light, fc.light_m, sizeof(xf.light));
    uint32_t* pair_count = f->shadow.ws.clip_count + 6;  // cleared with the pass's counters
#ifdef WR_SHADOW_SETUP_PER_FACE
    hipLaunchKernelGGL(k_setup<true>, grid1(g.ntris, 6), dim3(256), 0, st, g.pos, g.idx, g.ntris,
                       f->shadow.ws, ctx->S, ctx->S, 0, 1, xf);
    (void)pair_count;
#else
    hipLaunchKernelGGL(k_shadow_classify, grid1(g.ntris), dim3(256), 0, st, g.pos, g.idx, g.ntris, f->shadow.ws, xf, pair_count, f->d_shadow_pairs);
    hipLaunchKernelGGL(k_setup_pairs, grid1(std::max(2u * g.ntris, 256u)), dim3(256), 0, st, g.pos, g.idx, f->shadow.ws, ctx->S, ctx->S, xf, pair_count,
                       f->d_shadow_pairs);
#endif
    hipLaunchKernelGGL(k_setup_clip<true>, dim3(128, 6, 1), dim3(64), 0, st, g.pos, g.idx, f->shadow.ws,
                       ctx->S, ctx->S, 0, 1, xf);
    return bin_and_raster(ctx, f, st, f->shadow, true, 0, 1, prev_done);
}

// World-space box around the points the fog marches of this frame can sample: the part of the
// fog model's bounding box surface that faces the camera and lies in the view frustum (where
// fog fragments, i.e. march origins, can be), each point extended by the longest view march
// (30 steps of 5 units along its view ray), the whole then extended towards the light by at
// most the longest light march (10 steps of 10 units), plus a margin.  An estimate is enough:
// a sample outside a noise table is evaluated directly, with the same result.
bool fog_sample_bounds(const WrContext* ctx, const FrameConsts& fc, double lo[3], double hi[3]) {
    const float* a = ctx->fog_aabb;
    double corner[8][3];
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {a[(c & 1) ? 3 : 0], a[(c & 2) ? 4 : 1], a[(c & 4) ? 5 : 2]};
        for (int r = 0; r < 3; ++r) {
            corner[c][r] = fc.model_fog[r] * p[0] + fc.model_fog[4 + r] * p[1] + fc.model_fog[8 + r] * p[2] + fc.model_fog[12 + r];
            if (!(std::fabs(corner[c][r]) < 1e7)) return false;
        }
    }
    static const int faces[6][4] = {{0, 2, 6, 4}, {1, 5, 7, 3}, {0, 4, 5, 1}, {2, 3, 7, 6}, {0, 1, 3, 2}, {4, 6, 7, 5}};
    double centre[3] = {0, 0, 0};
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 3; ++r) centre[r] += corner[c][r] / 8.0;
    for (int r = 0; r < 3; ++r) { lo[r] = 1e300; hi[r] = -1e300; }
    const double view_reach = (double)kFogMaxSteps * kFogStepSize;
    const double light_reach = (double)kFogLightMaxSteps * kFogLightStepSize;
    bool any = false;
    for (int fi = 0; fi < 6; ++fi) {
        // faces turned away from the camera produce no fragments (back-face culling); which way
        // a face looks is judged from the box centre, so the vertex order above does not matter
        double fc_[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 3; ++r) fc_[r] += corner[faces[fi][k]][r] / 4.0;
        double out_dot = 0.0;
        for (int r = 0; r < 3; ++r) out_dot += (fc_[r] - centre[r]) * ((double)fc.cam_pos[r] - fc_[r]);
        if (!(out_dot > 0.0)) continue;
        // clip the face against the six frustum planes in clip space, carrying world positions
        struct P { double w[3]; double c[4]; };
        std::vector<P> poly, next;
        for (int k = 0; k < 4; ++k) {
            P q;
            for (int r = 0; r < 3; ++r) q.w[r] = corner[faces[fi][k]][r];
            for (int r = 0; r < 4; ++r) q.c[r] = fc.pv[r] * q.w[0] + fc.pv[4 + r] * q.w[1] + fc.pv[8 + r] * q.w[2] + fc.pv[12 + r];
            poly.push_back(q);
        }
        for (int pl = 0; pl < 6 && !poly.empty(); ++pl) {
            auto dist = [pl](const P& q) {
                switch (pl) {
                    case 0: return q.c[3] + q.c[0];
                    case 1: return q.c[3] - q.c[0];
                    case 2: return q.c[3] + q.c[1];
                    case 3: return q.c[3] - q.c[1];
                    case 4: return q.c[2];
                    default: return q.c[3] - q.c[2];
                }
            };
            next.clear();
            for (size_t k = 0; k < poly.size(); ++k) {
                const P& p0 = poly[k];
                const P& p1 = poly[(k + 1) % poly.size()];
                const double d0 = dist(p0), d1 = dist(p1);
                if (d0 >= 0.0) next.push_back(p0);
                if ((d0 >= 0.0) != (d1 >= 0.0)) {
                    const double t = d0 / (d0 - d1);
                    P q;
                    for (int r = 0; r < 3; ++r) q.w[r] = p0.w[r] + t * (p1.w[r] - p0.w[r]);
                    for (int r = 0; r < 4; ++r) q.c[r] = p0.c[r] + t * (p1.c[r] - p0.c[r]);
                    next.push_back(q);
                }
            }
            poly.swap(next);
        }
        for (const P& q : poly) {
            double d[3], len = 0.0;
            for (int r = 0; r < 3; ++r) { d[r] = q.w[r] - (double)fc.cam_pos[r]; len += d[r] * d[r]; }
            len = std::sqrt(len);
            for (int r = 0; r < 3; ++r) {
                const double e = len > 0.0 ? q.w[r] + view_reach * d[r] / len : q.w[r];
                lo[r] = std::min(lo[r], std::min(q.w[r], e));
                hi[r] = std::max(hi[r], std::max(q.w[r], e));
            }
            any = true;
        }
    }
    if (!any) return false;
    for (int r = 0; r < 3; ++r) {
        const double L = fc.light_pos[r];
        lo[r] = std::min(lo[r], std::max(L, lo[r] - light_reach)) - 3.0;
        hi[r] = std::max(hi[r], std::min(L, hi[r] + light_reach)) + 3.0;
    }
    return true;
}

// Chooses which noise levels get a hash table this frame.  A level whose lattice over the whole
// box of fog_sample_bounds fits the per-level and pool budgets is tabled whole.  Of the levels
// that do not fit, the coarsest ones get a table over a part of the box, as long as the pool has
// room: the part within a distance R of the camera along every axis, R as large as the budget
// allows.  Samples are dense where the rays are close together, near the camera (half of the fog
// pixels of the reference's view start their march within 400 units of it), so a table that
// covers a small share of the volume serves a large share of the samples; a sample outside a
// table is evaluated directly, with the same result.
#ifndef WR_NOISE_PARTIAL
#define WR_NOISE_PARTIAL 1
#endif
#ifndef WR_NOISE_PARTIAL_MIN_ENTRIES
#define WR_NOISE_PARTIAL_MIN_ENTRIES (1u << 18)
#endif
NoiseTables plan_noise_tables(const WrContext* ctx, const FrameRes* f, const FrameConsts& fc, uint32_t* max_entries) {
    NoiseTables nt;
    memset(&nt, 0, sizeof(nt));
    noise_tables_clear(nt);
    *max_entries = 0;
    double lo[3], hi[3];
    if (!fog_sample_bounds(ctx, fc, lo, hi)) return nt;
    const double drift[3] = {fc.drift_xy, fc.drift_xy, fc.drift_z};
    size_t used = 0;
    // lattice box [l, l + n) of level (k, o) over the world-space box [blo, bhi]
    auto lattice = [&](int k, int o, const double blo[3], const double bhi[3], long l[3], long n[3]) {
        const double s = (k == 0 ? 0.01 : 0.05), m = (double)(1 << o);
        for (int r = 0; r < 3; ++r) {
            const double a = (blo[r] * s + drift[r]) * m, b = (bhi[r] * s + drift[r]) * m;
            l[r] = (long)std::floor(a) - 1;
            n[r] = (long)std::floor(b) + 2 - l[r] + 1;
        }
        return (double)n[0] * n[1] * n[2];
    };
    auto take = [&](int k, int o, const long l[3], const long n[3], double entries) {
        NoiseTable& t = nt.level[k * 8 + o];
        t.data = f->d_noise_pool + used;
        t.lox = (int)l[0]; t.loy = (int)l[1]; t.loz = (int)l[2];
        t.nx = (int)n[0]; t.ny = (int)n[1]; t.nz = (int)n[2];
        noise_table_seal(t);
        used += (size_t)entries;
        *max_entries = std::max<uint32_t>(*max_entries, (uint32_t)entries);
    };
    const double level_cap = (double)std::min(kNoiseLevelEntries, f->noise_pool_entries);
    bool whole[16];
    for (int k = 0; k < 2; ++k)
        for (int o = 0; o < 8; ++o) {
            long l[3], n[3];
            const double entries = lattice(k, o, lo, hi, l, n);
#ifdef WR_DEV
            if (getenv("WR_DEBUG_TABLES")) fprintf(stderr, "noise level %d/%d: %ld x %ld x %ld = %.0f entries\n", k, o, n[0], n[1], n[2], entries);
#endif
            whole[k * 8 + o] = !(entries > level_cap || used + (size_t)entries > f->noise_pool_entries);
            if (whole[k * 8 + o]) take(k, o, l, n, entries);
        }
    if (!WR_NOISE_PARTIAL) return nt;
    // Partial tables, coarsest cells first (cell size of level (k, o): 100 / 2^o and 20 / 2^o
    // units).  The second fbm's cells count double: its octaves run for about nine samples in
    // ten, the first fbm's fine octaves for four in ten (its coarse octaves decide most samples).
    // A part that reaches less than 100 units into the box is not worth its entries.
    double reach_needed = 0.0;
    for (int r = 0; r < 3; ++r)
        reach_needed = std::max(reach_needed, std::max(lo[r] - (double)fc.cam_pos[r], (double)fc.cam_pos[r] - hi[r]));
    reach_needed = std::max(reach_needed, 0.0) + 100.0;
    for (int rank = 0; rank < 16; ++rank) {
        int best = -1;
        double best_cell = 0.0;
        for (int l = 0; l < 16; ++l) {
            const double cell = (l < 8 ? 100.0 : 40.0) / (double)(1 << (l & 7));
            if (!whole[l] && nt.level[l].data == nullptr && cell > best_cell) { best = l; best_cell = cell; }
        }
        if (best < 0) break;
        whole[best] = true;  // (looked at)
        const int k = best >> 3, o = best & 7;
        const double budget = std::min(level_cap, (double)(f->noise_pool_entries - used));
        if (budget < (double)WR_NOISE_PARTIAL_MIN_ENTRIES) break;
        // the largest R whose box (the bounds within R of the camera along every axis) fits the budget
        auto box_of = [&](double R, double blo[3], double bhi[3]) {
            for (int r = 0; r < 3; ++r) {
                blo[r] = std::max(lo[r], (double)fc.cam_pos[r] - R);
                bhi[r] = std::min(hi[r], (double)fc.cam_pos[r] + R);
                if (!(blo[r] <= bhi[r])) return false;
            }
            return true;
        };
        double r_ok = 0.0, r_lo = 0.0, r_hi = 1.0e5;
        for (int it = 0; it < 40; ++it) {
            const double R = 0.5 * (r_lo + r_hi);
            double blo[3], bhi[3];
            long l[3], n[3];
            if (!box_of(R, blo, bhi) || lattice(k, o, blo, bhi, l, n) <= budget) { r_ok = R; r_lo = R; } else r_hi = R;
        }
        double blo[3], bhi[3];
        long l[3], n[3];
        if (r_ok < reach_needed || !box_of(r_ok, blo, bhi)) continue;
        const double entries = lattice(k, o, blo, bhi, l, n);
        if (entries > budget || n[0] < 3 || n[1] < 3 || n[2] < 3) continue;
#ifdef WR_DEV
        if (getenv("WR_DEBUG_TABLES")) fprintf(stderr, "noise level %d/%d: part within %.0f units of the camera, %ld x %ld x %ld = %.0f entries\n", k, o, r_ok, n[0], n[1], n[2], entries);
#endif
        take(k, o, l, n, entries);
    }
    return nt;
}

// One frame.  Main stream: geometry pass, shading (after the shadow maps), fog blend.  Side
// stream: the six shadow passes.  The fog pass has two schedules:
//  * one frame in flight (the frame has the device to itself): a third stream takes the fog
//    volume set-up and the noise tables (they need the frame constants only), then, once the
//    geometry depth exists, the view march (k_fog_noise_view: depth and tables, no shadow
//    maps), and after the shadow maps the fog points' shadow lookups and the light march.  The
//    dominant kernel starts as soon as the geometry raster is done, runs next to the shadow
//    passes and the shading kernels, and its evenly sized work units fill the chip to its end;
//  * several frames in flight: the fog pass follows the shadow passes on the side stream and
//    marches each item in one piece (k_fog_march); the neighbouring frames' kernels fill the
//    device around it (measured on one machine: 968 against 889 frames/s with four frames in
//    flight, 757 against 782 with one).
// Profiling events of a frame: [0] frame start, [3] geometry end, [4] shade end, [7] frame end
// on the main stream; [1] shadow start, [2] shadow end on the side stream; [8] fog start,
// [6] view-march noise (or k_fog_march) start, [9] its end, [10] light-march noise start,
// [5] its end on the fog stream.
int frame_stages(WrContext* ctx, FrameRes* f, const FrameConsts& fc, uint32_t passes, bool do_shadow) {
    const ModelDev& g = ctx->dev[WR_MODEL_GEOMETRY];
    const ModelDev& lm = ctx->dev[WR_MODEL_LIGHT_MARKER];
    const ModelDev& fm = ctx->dev[WR_MODEL_FOG_VOLUME];
    const bool prof = ctx->profiling == 1;        // all stage events
    // The event pair around the dominant kernel.  Every event on the fog stream is on the frame's
    // longest chain (about 1 % of a frame each with one frame in flight): level 2 records these two
    // only, and with one frame in flight only for every fourth frame.
    const bool prof_march = ctx->profiling == 1 || (ctx->profiling == 2 && (ctx->nfr > 1 || (ctx->prof_tick++ % kProfEveryAlone) == 0));
    hipEvent_t* ev = ctx->ev[ctx->ev_next];
    static const char* sched_env = getenv("WR_DEBUG_SCHEDULE");  // "split" / "mono": force a fog schedule
    const bool latency_schedule = sched_env ? sched_env[0] == 's' : ctx->nfr == 1;
    hipStream_t ms = f->ms, ss = f->side, fs = latency_schedule ? f->fs : f->side;
    // One frame in flight: a frame ends with the fog pass's tail (light march, fog blend) on its
    // main stream, while its other stream is idle since the shading kernels ended 0.1 ms earlier;
    // the next frame begins with 0.13 ms of vertex and triangle set-up that leave most of the
    // device idle and touch nothing the fog tail reads.  So the two streams exchange roles with
    // every frame: the next frame's geometry set-up queues behind the previous frame's shading
    // (whose inputs it overwrites) and runs next to its fog tail; what does touch the tail's data
    // is behind it on the tail's stream (fog volume, noise tables, later the shading kernels) or
    // waits for the previous frame's end (the two rasterisers: depth and shadow maps).
    const bool flipping = latency_schedule && f->fs != f->ms && f->fs != f->side && f->side != f->ms;
    if (flipping) {
        if (ctx->flip_streams) std::swap(ms, fs);
        ctx->flip_streams = !ctx->flip_streams;
    }
    f->cur_ms = ms;
    const hipEvent_t prev_done = flipping ? f->ev_targets_free : nullptr;
    int rc;
    // Order the frame after what has been queued on the context's stream so far.  With one frame
    // in flight, a stream nobody else can queue on (the library's own) and nothing queued there
    // by the library since the last frame, the only thing on it is the wait for that frame, which
    // this frame's main stream is behind anyway: the hop to that stream and back (two event
    // waits, about 25 us of an interactive frame) is left out.
    const bool follows_own_frame = ctx->nfr == 1 && ctx->own_stream && !ctx->stream_has_work && f == &ctx->fr[0];
    if (!follows_own_frame) {
        WR_HIP(hipEventRecord(f->ev_fork, ctx->stream));
        WR_HIP(hipStreamWaitEvent(ms, f->ev_fork, 0));
    }
    ctx->stream_has_work = false;
    if (prof) WR_HIP(hipEventRecord(ev[0], ms));
    const bool fog_pass = (passes & WR_PASS_FOG) && fm.ntris;

    // Streams exchanging roles and nothing of the caller's to wait for: the shadow passes' set-up
    // touches only what the previous frame's shadow passes, earlier on the same stream, used, so it
    // need not wait for this frame's start, which is the end of the previous frame's shading.  It is
    // let go when that shading is down to its last, short kernel, and saves the 20 us an event
    // takes to cross streams.  (Let go earlier, before the ordered blending or further back, it
    // runs in the middle of the previous frame and takes the slots of that frame's critical
    // kernels: 930-965 against 1030 frames/s.)  Its counters are then cleared on its own stream.
    const bool early_shadows = flipping && follows_own_frame;
    // every counter block of the frame's three raster work spaces and of the ordered blending, in
    // one launch, before the streams part
    {
        ClearList cl{};
        cl.ptr[0] = f->geom.counters; cl.words[0] = (uint32_t)(f->geom.counters_bytes / 4);
        cl.ptr[1] = f->blend.counts; cl.words[1] = (uint32_t)(f->blend_clear_bytes / 4);
        if (!early_shadows) { cl.ptr[2] = f->shadow.counters; cl.words[2] = (uint32_t)(f->shadow.counters_bytes / 4); }
        if (!flipping) { cl.ptr[3] = f->fog.counters; cl.words[3] = (uint32_t)(f->fog.counters_bytes / 4); }
        hipLaunchKernelGGL(k_clear_words, dim3(32), dim3(256), 0, ms, cl);
    }
    // after this set's previous frame and the caller's earlier work
    WR_HIP(hipEventRecord(f->ev_start, ms));
    if (early_shadows) {
        WR_HIP(hipStreamWaitEvent(ss, f->ev_shade_tail, 0));  // (the previous frame's: this frame records it further down)
        ClearList cs{};
        cs.ptr[0] = f->shadow.counters; cs.words[0] = (uint32_t)(f->shadow.counters_bytes / 4);
        hipLaunchKernelGGL(k_clear_words, dim3(8), dim3(256), 0, ss, cs);
    } else {
        WR_HIP(hipStreamWaitEvent(ss, f->ev_start, 0));
    }
    if (fs != ss) WR_HIP(hipStreamWaitEvent(fs, f->ev_start, 0));
    if (flipping) {
        // (the previous frame's fog tail reads the fog pass's counters: cleared behind it)
        ClearList cg{};
        cg.ptr[0] = f->fog.counters; cg.words[0] = (uint32_t)(f->fog.counters_bytes / 4);
        hipLaunchKernelGGL(k_clear_words, dim3(8), dim3(256), 0, fs, cg);
    }
    FogArgs fa;
    memset(&fa, 0, sizeof(fa));
    auto fill_fog_args = [&]() {
        fa.fc = fc; fa.ws = f->fog.ws; fa.pos = fm.pos; fa.idx = fm.idx; fa.ntris = fm.ntris;
        fa.depth = f->d_depth; fa.color = f->d_color; fa.shadow = f->d_shadow; fa.tables = ctx->d_tables;
        fa.counters = ctx->d_counters;
        // the fog workspace has one bin and is not binned, so its bin_count and bin_cursor words
        // are free: they count the march items and the pixels with several fog fragments; the
        // word after its clip_count counts the lit fog points
        fa.item_count = f->fog.ws.bin_count; fa.item_pixel = f->d_fog_pixel; fa.item_ray = f->d_fog_ray;
        fa.multi_count = f->fog.ws.bin_cursor; fa.multi_pixel = f->d_fog_multi;
        fa.lit_count = f->fog.ws.clip_count + 1; fa.lit_item = f->d_fog_lit;
        fa.item_res = f->d_fog_res; fa.noise_view = f->d_fog_noise_view; fa.noise_light = f->d_fog_noise_light;
        fa.cap = (uint32_t)((size_t)ctx->W * ctx->H);
    };
    if (fog_pass) fill_fog_args();
    const dim3 fog_grid((uint32_t)(ctx->W + kFogBlock - 1) / kFogBlock, (uint32_t)(ctx->H + kFogBlock - 1) / kFogBlock, 1);
    auto submit_fog_volume = [&]() {
        hipLaunchKernelGGL(k_setup_small, dim3((fm.ntris + kSmallTris - 1) / kSmallTris), dim3(64), 0, fs, fm.pos, fm.idx, fm.ntris, 1, 1, 0u, f->fog.ws, fc);
    };
    // Submission takes the host about as long as the device needs for half a frame, so the
    // order of the two independent chains matters: the one the frame's critical path starts
    // with goes first (one frame in flight: geometry, which the view march waits for; frames
    // overlapping: the shadow passes, the longer chain, which shading waits for).
    auto submit_geometry = [&]() -> int {
        // main stream: geometry + light marker -> visibility / depth
        VertexIn vin{g.pos, g.uv, g.nrm, g.tan, g.bit, g.nv};
        hipLaunchKernelGGL(k_vs_geometry, grid1(g.nv), dim3(256), 0, ms, vin, f->vout, fc);
        hipLaunchKernelGGL(k_setup<false>, grid1(g.ntris, 1), dim3(256), 0, ms, f->vout.clip, g.idx, g.ntris,
                           f->geom.ws, ctx->W, ctx->H, fc.geom_tile_rank, fc.geom_tile_world, ShadowXf{});
        hipLaunchKernelGGL(k_setup_clip<false>, dim3(256, 1, 1), dim3(64), 0, ms, f->vout.clip, g.idx, f->geom.ws,
                           ctx->W, ctx->H, fc.geom_tile_rank, fc.geom_tile_world, ShadowXf{});
        if (lm.ntris)
            hipLaunchKernelGGL(k_setup_small, dim3((lm.ntris + kSmallTris - 1) / kSmallTris), dim3(64), 0, ms, lm.pos, lm.idx, lm.ntris, 0,
                               (passes & WR_PASS_LIGHT_DEBUG) ? 1 : 0, g.ntris, f->geom.ws, fc);
        if ((rc = bin_and_raster(ctx, f, ms, f->geom, false, fc.geom_tile_rank, fc.geom_tile_world, prev_done)) != WR_OK) return rc;
        if (prof) WR_HIP(hipEventRecord(ev[3], ms));
        WR_HIP(hipEventRecord(f->ev_geom_done, ms));
        return WR_OK;
    };
    auto submit_shadows = [&]() -> int {
        // side stream: shadow maps
        if (prof) WR_HIP(hipEventRecord(ev[1], ss));
        if (do_shadow && (passes & WR_PASS_SHADOWS)) {
            if ((rc = shadow_stage(ctx, f, ss, fc, prev_done)) != WR_OK) return rc;
            f->shadow_is_clear = false;
        } else if (do_shadow && !f->shadow_is_clear) {
            // No shadow passes in this frame: it sees cleared maps, whatever frame used this set
            // of buffers before (the reference clears each face to 1.0 before drawing into it,
            // state.rs:655-657; a frame that draws nothing leaves that).
            const size_t ns = shadow_buffer_floats(ctx->S);
            if (prev_done) WR_HIP(hipStreamWaitEvent(ss, prev_done, 0));
            hipLaunchKernelGGL(k_fill_f32, dim3((uint32_t)((ns + 255) / 256)), dim3(256), 0, ss, f->d_shadow, ns, 1.0f);
            f->shadow_is_clear = true;
        }
        if (prof) WR_HIP(hipEventRecord(ev[2], ss));
        WR_HIP(hipEventRecord(f->ev_shadow_done, ss));
        return WR_OK;
    };
    // fog volume set-up and noise tables: they need the frame constants only
    auto submit_fog_prologue = [&]() -> int {
        if (fog_pass) {
            submit_fog_volume();
            if (fs != ms) WR_HIP(hipEventRecord(f->ev_fog_volume_done, fs));
            uint32_t max_entries = 0;
            fa.tables_noise = plan_noise_tables(ctx, f, fc, &max_entries);
            if (max_entries) {
                TablePlan plan{};
                for (int l = 0; l < 16; ++l) {
                    const NoiseTable& t = fa.tables_noise.level[l];
                    plan.xchunks[l] = plan.ychunks[l] = 1u;
                    plan.wg_begin[l + 1] = plan.wg_begin[l];
                    if (t.data) plan_table_level(plan, l, t);
                }
                hipLaunchKernelGGL(k_fog_tables, dim3(plan.wg_begin[16]), dim3(256), 0, fs, fa.tables_noise, plan);
            }
        }
        return WR_OK;
    };
    if (latency_schedule) {
        if ((rc = submit_geometry()) != WR_OK) return rc;
        if ((rc = submit_fog_prologue()) != WR_OK) return rc;
        if ((rc = submit_shadows()) != WR_OK) return rc;
    } else {
        if ((rc = submit_shadows()) != WR_OK) return rc;
        if ((rc = submit_geometry()) != WR_OK) return rc;
        if ((rc = submit_fog_prologue()) != WR_OK) return rc;
    }
    // One frame in flight: the frame's critical path is geometry -> fog set-up -> marches -> fog
    // blend, so those stay on ONE stream (the main one; a wait across streams costs 10-25 us each
    // time) and the shading kernels, which have slack, move to the fog stream once it has done the
    // fog volume set-up and the tables.  Frames overlapping: the fog pass follows the shadow
    // passes on the side stream, shading stays on the main stream.
    const bool swap_roles = latency_schedule && fs != ms;
    hipStream_t fog_st = swap_roles ? ms : fs, shade_st = swap_roles ? fs : ms;
    if (swap_roles) {
        WR_HIP(hipEventRecord(f->ev_prologue_done, fs));
        if (fog_pass) WR_HIP(hipStreamWaitEvent(ms, f->ev_fog_volume_done, 0));
        WR_HIP(hipStreamWaitEvent(fs, f->ev_geom_done, 0));
    } else {
        WR_HIP(hipStreamWaitEvent(fs, f->ev_geom_done, 0));
    }
    if (prof) WR_HIP(hipEventRecord(ev[8], fog_st));
    const size_t npx = (size_t)ctx->W * ctx->H;
    const uint32_t item_groups = (uint32_t)((npx + 255) / 256);
    if (fog_pass) {
        if (ctx->counting) hipLaunchKernelGGL(k_fog_setup<true>, fog_grid, dim3(kFogSetupThreads), 0, fog_st, fa);
        else hipLaunchKernelGGL(k_fog_setup<false>, fog_grid, dim3(kFogSetupThreads), 0, fog_st, fa);
        if (swap_roles) WR_HIP(hipStreamWaitEvent(ms, f->ev_prologue_done, 0));  // the noise tables
        if (prof_march) WR_HIP(hipEventRecord(ev[6], fog_st));
        if (latency_schedule) {
            const dim3 vg(item_groups * (uint32_t)kFogViewGroups, 1, 1);
            if (ctx->counting) hipLaunchKernelGGL(k_fog_noise_view<true>, vg, dim3(256), 0, fog_st, fa);
            else hipLaunchKernelGGL(k_fog_noise_view<false>, vg, dim3(256), 0, fog_st, fa);
            if (prof_march) WR_HIP(hipEventRecord(ev[9], fog_st));
            WR_HIP(hipStreamWaitEvent(fog_st, f->ev_shadow_done, 0));
            if (ctx->counting) hipLaunchKernelGGL(k_fog_view_resolve<true>, dim3((item_groups + kResolveChunks - 1) / kResolveChunks), dim3(256), 0, fog_st, fa);
            else hipLaunchKernelGGL(k_fog_view_resolve<false>, dim3((item_groups + kResolveChunks - 1) / kResolveChunks), dim3(256), 0, fog_st, fa);
            if (prof) WR_HIP(hipEventRecord(ev[10], fog_st));
#ifdef WR_DEV
            for (int l = 0; l < 16; ++l) fa.tables_noise.level[l].dbg_level = 256 + l;
#endif
            const dim3 lg(item_groups * (uint32_t)kFogLightGroups, 1, 1);
            if (ctx->counting) hipLaunchKernelGGL(k_fog_noise_light<true>, lg, dim3(256), 0, fog_st, fa);
            else hipLaunchKernelGGL(k_fog_noise_light<false>, lg, dim3(256), 0, fog_st, fa);
        } else {
            // (the side stream has the shadow maps: it rendered them)
            const dim3 mg((uint32_t)((npx + 63) / 64), 1, 1);
            if (ctx->counting) hipLaunchKernelGGL(k_fog_march<true>, mg, dim3(64), 0, fog_st, fa);
            else hipLaunchKernelGGL(k_fog_march<false>, mg, dim3(64), 0, fog_st, fa);
            if (prof_march) WR_HIP(hipEventRecord(ev[9], fog_st));
            if (prof) WR_HIP(hipEventRecord(ev[10], fog_st));
        }
    } else if (prof_march) {
        WR_HIP(hipEventRecord(ev[6], fog_st));
        WR_HIP(hipEventRecord(ev[9], fog_st));
        if (prof) WR_HIP(hipEventRecord(ev[10], fog_st));
    }
    if (prof) WR_HIP(hipEventRecord(ev[5], fog_st));
    if (!swap_roles) WR_HIP(hipEventRecord(f->ev_join, fog_st));  // what the main stream waits for before the fog blend:

    // shading (reads the shadow maps; on another stream than the geometry pass it also waits for that)
    WR_HIP(hipStreamWaitEvent(shade_st, f->ev_shadow_done, 0));
    ShadeArgs sa;
    sa.fc = fc; sa.ws = f->geom.ws; sa.vis = f->d_vis; sa.depth = f->d_depth; sa.color = f->d_color;
    sa.idx = g.idx; sa.tri_mat = g.tri_mat; sa.ntris_geom = g.ntris;
    sa.v0 = f->vout.v0; sa.v1 = f->vout.v1; sa.v2 = f->vout.v2; sa.v3 = f->vout.v3;
    sa.mats = ctx->d_mats; sa.tex = ctx->d_tex; sa.texels = ctx->d_texels; sa.tables = ctx->d_tables;
    sa.shadow = f->d_shadow; sa.counters = ctx->d_counters;
    sa.bw = f->blend;
    dim3 pg((uint32_t)(ctx->W + 15) / 16, (uint32_t)(ctx->H + 15) / 16, 1);
    if (ctx->counting) hipLaunchKernelGGL(k_shade<true>, pg, dim3(256), 0, shade_st, sa);
    else hipLaunchKernelGGL(k_shade<false>, pg, dim3(256), 0, shade_st, sa);
    // ordered blending of the pixels k_shade flagged: collect, order, shade, blend
    {
        const RasterWS& ws = f->geom.ws;
        LayerSink ls{f->blend.rp_slot, f->d_vis, f->blend.bin_flag, f->blend.lists, f->blend.list_count, f->blend.cap};
        // (a frame that has the device to itself: the slowest bins, 140 us each with 256 threads,
        // are the length of this launch, and a whole compute unit's worth of threads shortens them;
        // frames overlapping: small workgroups find room sooner, 1111 against 1060 frames/s)
        if (ctx->nfr == 1)
            hipLaunchKernelGGL((k_raster<kRasterLayers, kLayerRasterThreadsAlone>), dim3((uint32_t)(ws.bins_x * ws.bins_y), 1, 1), dim3(kLayerRasterThreadsAlone), 0,
                               shade_st, ws, ctx->W, ctx->H, (uint32_t*)nullptr, (float*)nullptr, fc.tile_rank, fc.tile_world, ls);
        else
            hipLaunchKernelGGL((k_raster<kRasterLayers, kColorRasterThreads>), dim3((uint32_t)(ws.bins_x * ws.bins_y), 1, 1), dim3(kColorRasterThreads), 0,
                               shade_st, ws, ctx->W, ctx->H, (uint32_t*)nullptr, (float*)nullptr, fc.tile_rank, fc.tile_world, ls);
        const uint32_t slot_groups = (uint32_t)std::min<size_t>((npx + 255) / 256, 2048);
        hipLaunchKernelGGL(k_resolve_shade, dim3(slot_groups), dim3(256), 0, shade_st, sa);
        if (flipping) WR_HIP(hipEventRecord(f->ev_shade_tail, shade_st));
        if (ctx->counting) hipLaunchKernelGGL(k_replay_slow<true>, dim3(256), dim3(64), 0, shade_st, sa);
        else hipLaunchKernelGGL(k_replay_slow<false>, dim3(256), dim3(64), 0, shade_st, sa);
    }
    if (prof) WR_HIP(hipEventRecord(ev[4], shade_st));
    if (swap_roles) WR_HIP(hipEventRecord(f->ev_join, shade_st));  // the fog pass, or (roles swapped) the shading

    // main stream: fog colours over the finished geometry colour
    WR_HIP(hipStreamWaitEvent(ms, f->ev_join, 0));
    if (fog_pass) {
        // (the two kernels write different pixels: a pixel with one fog fragment is a march item,
        // one with several is on k_fog_multi's list.  k_fog_multi is the last reader of the depth,
        // the shadow maps and the fog volume's records: the next frame's rasterisers wait for it,
        // not for the frame's end.)
        if (latency_schedule) {
            if (ctx->counting) hipLaunchKernelGGL(k_fog_multi<true>, dim3(1024), dim3(64), 0, ms, fa);
            else hipLaunchKernelGGL(k_fog_multi<false>, dim3(1024), dim3(64), 0, ms, fa);
            WR_HIP(hipEventRecord(f->ev_targets_free, ms));
            if (ctx->counting) hipLaunchKernelGGL(k_fog_shade_blend<true>, dim3(2048), dim3(256), 0, ms, fa);
            else hipLaunchKernelGGL(k_fog_shade_blend<false>, dim3(2048), dim3(256), 0, ms, fa);
        } else {
            hipLaunchKernelGGL(k_fog_blend, dim3(2048), dim3(256), 0, ms, fa);
            if (ctx->counting) hipLaunchKernelGGL(k_fog_multi<true>, dim3(1024), dim3(64), 0, ms, fa);
            else hipLaunchKernelGGL(k_fog_multi<false>, dim3(1024), dim3(64), 0, ms, fa);
        }
    } else if (latency_schedule) {
        WR_HIP(hipEventRecord(f->ev_targets_free, ms));
    }
    if (ctx->cfg.color_format == WR_FORMAT_BGRA8_UNORM_SRGB)
        hipLaunchKernelGGL(k_swap_red_blue, dim3(1024), dim3(256), 0, ms, f->d_color, npx);
    if (prof) WR_HIP(hipEventRecord(ev[7], ms));
    if (prof_march) {
        ctx->ev_next = (ctx->ev_next + 1) % WrContext::kEvFrames;
        if (ctx->ev_count < WrContext::kEvFrames) ++ctx->ev_count;
    }
    WR_HIP(hipGetLastError());
    return WR_OK;
}

// Marks the end of a frame's work on its main stream.  With one frame in flight the caller's
// stream continues after the frame, as if the frame had run there.
int finish_frame(WrContext* ctx, FrameRes* f) {
    WR_HIP(hipEventRecord(f->ev_done, f->cur_ms));
    f->submitted = true;
    f->serial = ++ctx->frames_submitted;
    if (ctx->nfr == 1) WR_HIP(hipStreamWaitEvent(ctx->stream, f->ev_done, 0));
    return WR_OK;
}

// Counting mode.  The counters are one block for all sets of frame buffers, so a counted call
// starts after every frame still in flight: its counters are its own.
int begin_counted(WrContext* ctx) {
    if (!ctx->counting) return WR_OK;
    for (int k = 0; k < ctx->nfr; ++k)
        if (ctx->fr[k].submitted) WR_HIP(hipStreamWaitEvent(ctx->stream, ctx->fr[k].ev_done, 0));
    WR_HIP(hipMemsetAsync(ctx->d_counters, 0, 8 * sizeof(unsigned long long), ctx->stream));
    ctx->stream_has_work = true;
    return WR_OK;
}

// The set of frame buffers the next frame goes to.
FrameRes* next_frame(WrContext* ctx) {
    ctx->cur = (ctx->cur + 1) % ctx->nfr;
    return &ctx->fr[ctx->cur];
}

int check_ready(WrContext* ctx, uint32_t passes) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    if (ctx->broken) return fail(ctx, WR_ERR_OUT_OF_MEMORY, "the last wr_resize failed: resize again before rendering");
    if (ctx->host[WR_MODEL_GEOMETRY].idx.empty() || !ctx->host[WR_MODEL_GEOMETRY].has_inst)
        return fail(ctx, WR_ERR_NO_SCENE, "geometry model needs at least one mesh and an instance");
    if (ctx->host[WR_MODEL_GEOMETRY].mats.empty())
        return fail(ctx, WR_ERR_NO_SCENE, "geometry model has no material");
    if ((passes & WR_PASS_FOG) && (ctx->host[WR_MODEL_FOG_VOLUME].idx.empty() || !ctx->host[WR_MODEL_FOG_VOLUME].has_inst))
        return fail(ctx, WR_ERR_NO_SCENE, "fog pass needs the fog-volume model and its instance");
    if ((passes & WR_PASS_LIGHT_DEBUG) && ctx->host[WR_MODEL_LIGHT_MARKER].idx.empty())
        return fail(ctx, WR_ERR_NO_SCENE, "light-debug pass needs the light-marker model");
    WR_HIP(hipSetDevice(ctx->cfg.device));
    if (ctx->scene_dirty) return build_scene(ctx);
    return WR_OK;
}

}  // namespace

extern "C" {

int wr_create(const WrConfig* config, WrContext** out_ctx) {
    if (!config || !out_ctx) return WR_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    if (config->width == 0 || config->height == 0 || config->width > 7168 || config->height > 7168) return WR_ERR_INVALID_ARGUMENT;
    if (config->color_format != WR_FORMAT_RGBA8_UNORM_SRGB && config->color_format != WR_FORMAT_BGRA8_UNORM_SRGB) return WR_ERR_UNSUPPORTED;
    uint32_t S = config->shadow_map_size ? config->shadow_map_size : 2048u;
    if (S > 7168) return WR_ERR_INVALID_ARGUMENT;
    if (config->tile_world > 1 && config->tile_rank >= config->tile_world) return WR_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WR_ERR_DEVICE;
    if (config->device < 0 || config->device >= ndev) return WR_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(config->device) != hipSuccess) return WR_ERR_DEVICE;
    if (config->device < kMaxScratchDevices) {
        // this device's copy of the static work buffer (every device has its own, at its own address)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[config->device]) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return WR_ERR_DEVICE;
            if (hipEventCreateWithFlags(&g_static_done[config->device], hipEventDisableTiming) != hipSuccess) return WR_ERR_DEVICE;
            g_scratch_ptr[config->device] = static_cast<unsigned char*>(p);
        }
    }
    WrContext* ctx = new WrContext();
    ctx->cfg = *config;
    ctx->W = (int)config->width; ctx->H = (int)config->height; ctx->S = (int)S;
    int rc = WR_OK;
    auto body = [&]() -> int {
        if (config->stream) ctx->stream = (hipStream_t)config->stream;
        else { WR_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
        for (int k = 0; k < WrContext::kEvFrames; ++k)
            for (int i = 0; i < WrContext::kEvPerFrame; ++i) WR_HIP(hipEventCreate(&ctx->ev[k][i]));
        ctx->nfr = (int)std::min<uint32_t>(std::max<uint32_t>(config->frames_in_flight, 1u), (uint32_t)WrContext::kMaxFrames);
        // Profiling aid: WR_DEBUG_ONE_STREAM=1 puts all of a frame's kernels on one stream, so
        // that a kernel trace shows every kernel's duration with the device to itself.
        const char* os_env = getenv("WR_DEBUG_ONE_STREAM");
        const bool one_stream = os_env && os_env[0] == '1';
        for (int k = 0; k < ctx->nfr; ++k) {
            FrameRes* f = &ctx->fr[k];
            WR_HIP(hipStreamCreateWithFlags(&f->ms, hipStreamNonBlocking));
            if (one_stream) {
                f->side = f->fs = f->ms;
            } else if (os_env && os_env[0] == '2') {  // fog work on the shadow stream
                WR_HIP(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
                f->fs = f->side;
            } else {
                WR_HIP(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
                WR_HIP(hipStreamCreateWithFlags(&f->fs, hipStreamNonBlocking));
            }
            hipEvent_t* evs[] = {&f->ev_fork, &f->ev_start, &f->ev_join, &f->ev_shadow_done, &f->ev_geom_done, &f->ev_done, &f->ev_prologue_done, &f->ev_fog_volume_done, &f->ev_targets_free, &f->ev_shade_tail};
            for (hipEvent_t* e : evs) WR_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        float tables[768];
        build_tables(tables);
        WR_HIP(hipMalloc(&ctx->d_tables, sizeof(tables)));
        WR_HIP(hipMemcpy(ctx->d_tables, tables, sizeof(tables), hipMemcpyHostToDevice));
        WR_HIP(hipMalloc(&ctx->d_counters, 8 * sizeof(unsigned long long)));
        WR_HIP(hipMemset(ctx->d_counters, 0, 8 * sizeof(unsigned long long)));
        WR_HIP(hipMalloc(&ctx->d_err, 4 * (WrContext::kMaxFrames + 1) * sizeof(uint32_t)));
        WR_HIP(hipMemset(ctx->d_err, 0, 4 * (WrContext::kMaxFrames + 1) * sizeof(uint32_t)));
        size_t ns = shadow_buffer_floats((int)S);  // faces, then the depth range of their 8x8 tiles
        for (int k = 0; k < ctx->nfr; ++k) {
            FrameRes* f = &ctx->fr[k];
            WR_HIP(hipMalloc(&f->d_noise_pool, kNoisePoolEntries * sizeof(float)));
            f->noise_pool_entries = kNoisePoolEntries;
            WR_HIP(hipMalloc(&f->d_shadow, ns * 4));
            hipLaunchKernelGGL(k_fill_f32, dim3((uint32_t)((ns + 255) / 256)), dim3(256), 0, ctx->stream, f->d_shadow, ns, 1.0f);
        }
        WR_HIP(hipStreamSynchronize(ctx->stream));
        return alloc_targets(ctx);
    };
    rc = body();
    if (rc != WR_OK) { wr_destroy(ctx); return rc; }
    *out_ctx = ctx;
    return WR_OK;
}

void wr_destroy(WrContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    for (FrameRes& f : ctx->fr) {
        if (f.ms) (void)hipStreamSynchronize(f.ms);
        if (f.side) (void)hipStreamSynchronize(f.side);
        if (f.fs) (void)hipStreamSynchronize(f.fs);
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    free_targets(ctx);
    dfree(ctx->d_tables); dfree(ctx->d_counters); dfree(ctx->d_err); dfree(ctx->d_gather_send); dfree(ctx->d_gather_recv);
    for (int k = 0; k < WrContext::kEvFrames; ++k)
        for (int i = 0; i < WrContext::kEvPerFrame; ++i) if (ctx->ev[k][i]) (void)hipEventDestroy(ctx->ev[k][i]);
    for (FrameRes& f : ctx->fr) {
        dfree(f.d_shadow); dfree(f.d_noise_pool);
        if (f.ms) (void)hipStreamDestroy(f.ms);
        if (f.side && f.side != f.ms) (void)hipStreamDestroy(f.side);
        if (f.fs && f.fs != f.ms && f.fs != f.side) (void)hipStreamDestroy(f.fs);
        hipEvent_t evs[] = {f.ev_fork, f.ev_start, f.ev_join, f.ev_shadow_done, f.ev_geom_done, f.ev_done, f.ev_prologue_done, f.ev_fog_volume_done, f.ev_targets_free, f.ev_shade_tail};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int wr_resize(WrContext* ctx, uint32_t width, uint32_t height) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    if (width == 0 || height == 0) return WR_OK;  // state.rs:576 ignores empty sizes
    if (width > 7168 || height > 7168) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "target larger than 7168");
    WR_HIP(hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != WR_OK) return rcs;
    }
    ctx->W = (int)width; ctx->H = (int)height;
    ctx->cfg.width = width; ctx->cfg.height = height;
    ctx->scene_dirty = true;
    const int rc = alloc_targets(ctx);
    ctx->broken = rc != WR_OK;
    if (ctx->broken) free_targets(ctx);
    return rc;
}

const char* wr_last_error(const WrContext* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int wr_upload_texture(WrContext* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, int srgb, uint32_t* out_texture) {
    if (!ctx || !rgba8 || width == 0 || height == 0) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "bad texture");
    TexHost t;
    t.w = width; t.h = height; t.srgb = srgb;
    t.px.assign(rgba8, rgba8 + (size_t)width * height * 4);
    ctx->tex_host.push_back(std::move(t));
    if (out_texture) *out_texture = (uint32_t)ctx->tex_host.size() - 1;
    ctx->scene_dirty = true;
    return WR_OK;
}

int wr_upload_material(WrContext* ctx, uint32_t model, uint32_t tb, uint32_t tn, uint32_t tmr, float metallic, float roughness,
                       uint32_t* out_material) {
    if (!ctx || model > 2) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "bad model");
    size_t nt = ctx->tex_host.size();
    if (tb >= nt || tn >= nt || tmr >= nt) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "material names an unknown texture");
    MatDesc m{tb, tn, tmr, metallic, roughness, 0, 0, 0};
    ctx->host[model].mats.push_back(m);
    if (out_material) *out_material = (uint32_t)ctx->host[model].mats.size() - 1;
    ctx->scene_dirty = true;
    return WR_OK;
}

int wr_upload_mesh(WrContext* ctx, uint32_t model, const WrModelVertex* vertices, uint32_t vertex_count,
                   const uint32_t* indices, uint32_t index_count, uint32_t material) {
    if (!ctx || model > 2 || !vertices || !indices) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "bad mesh");
    ModelHost& h = ctx->host[model];
    if (model == WR_MODEL_GEOMETRY && material >= h.mats.size())
        return fail(ctx, WR_ERR_INVALID_ARGUMENT, "mesh names an unknown material");
    for (uint32_t i = 0; i < index_count; ++i)
        if (indices[i] >= vertex_count) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "index out of range");
    uint32_t base = (uint32_t)h.verts.size();
    h.verts.insert(h.verts.end(), vertices, vertices + vertex_count);
    for (uint32_t i = 0; i + 2 < index_count; i += 3) {  // draw_indexed on a TriangleList ignores a partial last triangle
        h.idx.push_back(base + indices[i]);
        h.idx.push_back(base + indices[i + 1]);
        h.idx.push_back(base + indices[i + 2]);
        h.tri_mat.push_back(material);
    }
    ctx->scene_dirty = true;
    return WR_OK;
}

int wr_set_instances(WrContext* ctx, uint32_t model, const WrInstanceRaw* instances, uint32_t count) {
    if (!ctx || model > 2 || !instances) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "bad instances");
    if (count != 1) return fail(ctx, WR_ERR_UNSUPPORTED, "exactly one instance per model is supported");
    ctx->host[model].inst = instances[0];
    ctx->host[model].has_inst = true;
    return WR_OK;
}

int wr_clear_scene(WrContext* ctx) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    for (int m = 0; m < 3; ++m) ctx->host[m] = ModelHost();
    ctx->tex_host.clear();
    ctx->scene_dirty = true;
    return WR_OK;
}

int wr_render_frame(WrContext* ctx, const WrCameraUniform* camera, const WrLightUniform* light,
                    const WrGlobalUniforms* globals, uint32_t passes) {
    if (!ctx || !camera || !light || !globals) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null uniform");
    int rc = check_ready(ctx, passes);
    if (rc != WR_OK) return rc;
    if ((rc = begin_counted(ctx)) != WR_OK) return rc;
    FrameConsts fc = make_consts(ctx, camera, light, globals);
    FrameRes* f = next_frame(ctx);
    if ((rc = frame_stages(ctx, f, fc, passes, true)) != WR_OK) return rc;
    return finish_frame(ctx, f);
}

// The work buffers of one frame, carved from `ar` (no allocation): shadow maps, noise tables,
// target-sized and scene-sized buffers.  The colour target is the caller's.
static int carve_static_frame(WrContext* ctx, Arena& ar, FrameRes* f) {
    WR_HIP(arena_take(ar, f->d_shadow, shadow_buffer_floats(ctx->S) * 4));
    WR_HIP(arena_take(ar, f->d_noise_pool, kStaticNoisePoolEntries * sizeof(float)));
    f->noise_pool_entries = kStaticNoisePoolEntries;
    int rc = alloc_frame_targets(ctx, ar, f, false);
    if (rc != WR_OK) return rc;
    return alloc_frame_scene(ctx, ar, f);
}

int wr_render_frame_static(WrContext* ctx, const WrCameraUniform* camera, const WrLightUniform* light,
                           const WrGlobalUniforms* globals, uint32_t passes, void* device_color) {
    if (!ctx || !camera || !light || !globals || !device_color) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null argument");
    if (ctx->scene_dirty)
        return fail(ctx, WR_ERR_NO_SCENE, "the scene is not resident on the device yet: the first wr_render_frame after an upload "
                                          "or a resize makes it so (that call allocates, this one never does)");
    int rc = check_ready(ctx, passes);
    if (rc != WR_OK) return rc;
    if (ctx->cfg.tile_world > 1) return fail(ctx, WR_ERR_UNSUPPORTED, "wr_render_frame_static renders whole frames only");
    if (ctx->cfg.device >= kMaxScratchDevices || !g_scratch_ptr[ctx->cfg.device])
        return fail(ctx, WR_ERR_UNSUPPORTED, "no static work buffer on this device");
    FrameRes sf;
    const FrameRes& own = ctx->fr[0];  // streams and events exist since wr_create
    sf.ms = own.ms; sf.side = own.side; sf.fs = own.fs;
    sf.ev_fork = own.ev_fork; sf.ev_start = own.ev_start; sf.ev_join = own.ev_join;
    sf.ev_shadow_done = own.ev_shadow_done; sf.ev_geom_done = own.ev_geom_done; sf.ev_done = own.ev_done;
    sf.ev_prologue_done = own.ev_prologue_done; sf.ev_fog_volume_done = own.ev_fog_volume_done; sf.ev_targets_free = own.ev_targets_free; sf.ev_shade_tail = own.ev_shade_tail;
    Arena sizing;
    sizing.dry = true;
    if ((rc = carve_static_frame(ctx, sizing, &sf)) != WR_OK) return rc;
    if (sizing.off > kStaticScratchBytes) {
        char msg[256];
        snprintf(msg, sizeof(msg), "this target and scene need %zu bytes of work buffers; the static buffer of wr_render_frame_static holds %zu",
                 sizing.off, kStaticScratchBytes);
        return fail(ctx, WR_ERR_CAPACITY, msg);
    }
    Arena ar;
    ar.base = g_scratch_ptr[ctx->cfg.device];
    ar.cap = kStaticScratchBytes;
    if ((rc = carve_static_frame(ctx, ar, &sf)) != WR_OK) return rc;
    sf.d_color = static_cast<uint32_t*>(device_color);
    // One static frame at a time per device, whichever context or host thread it comes from: the
    // array is touched only after the static frame before it has finished on the device.
    std::lock_guard<std::mutex> static_lock(g_scratch_mutex);
    const int dev = ctx->cfg.device;
    auto submit = [&]() -> int {
        if (g_static_recorded[dev]) WR_HIP(hipStreamWaitEvent(ctx->stream, g_static_done[dev], 0));
        // The buffer keeps what the last call left: clear what a frame reads before writing it.
        const size_t npx = (size_t)ctx->W * ctx->H;
        WR_HIP(hipMemsetAsync(sf.blend.rp_slot, 0, npx * 4, ctx->stream));
        ctx->stream_has_work = true;
        if (!(passes & WR_PASS_SHADOWS)) {
            const size_t ns = shadow_buffer_floats(ctx->S);
            hipLaunchKernelGGL(k_fill_f32, dim3((uint32_t)((ns + 255) / 256)), dim3(256), 0, ctx->stream, sf.d_shadow, ns, 1.0f);
        }
        int rcs = begin_counted(ctx);
        if (rcs != WR_OK) return rcs;
        FrameConsts fc = make_consts(ctx, camera, light, globals);
        if ((rcs = frame_stages(ctx, &sf, fc, passes, true)) != WR_OK) return rcs;
        WR_HIP(hipEventRecord(sf.ev_done, sf.cur_ms));
        ctx->fr[0].submitted = true;
        ++ctx->static_frames_submitted;
        WR_HIP(hipStreamWaitEvent(ctx->stream, sf.ev_done, 0));
        WR_HIP(hipEventRecord(g_static_done[dev], ctx->stream));
        g_static_recorded[dev] = true;
        ctx->stream_has_work = true;  // (the next frame must not touch the shared streams' buffers early)
        return WR_OK;
    };
    rc = submit();
    // A call that failed half way may have queued work on the array that no event covers: the
    // device is drained before the next caller gets the array.
    if (rc != WR_OK) (void)sync_all(ctx);
    return rc;
}

int wr_render_batch(WrContext* ctx, const WrCameraUniform* cameras, uint32_t count, const WrLightUniform* light,
                    const WrGlobalUniforms* globals, uint32_t passes, int shadows_every_frame, void* device_frames) {
    if (!ctx || (!cameras && count) || !light || !globals) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null uniform");
    int rc = check_ready(ctx, passes);
    if (rc != WR_OK) return rc;
    if (count == 0) return WR_OK;  // nothing to render: the most recent frame stays the one read back
    if ((rc = begin_counted(ctx)) != WR_OK) return rc;
    size_t frame_bytes = (size_t)ctx->W * ctx->H * 4;
    FrameRes* f = next_frame(ctx);
    for (uint32_t i = 0; i < count; ++i) {
        FrameConsts fc = make_consts(ctx, &cameras[i], light, globals);
        // a batch that reuses the first frame's shadow maps stays on one set of frame buffers
        if (shadows_every_frame && i > 0) f = next_frame(ctx);
        if ((rc = frame_stages(ctx, f, fc, passes, shadows_every_frame || i == 0)) != WR_OK) return rc;
        if (device_frames)
            WR_HIP(hipMemcpyAsync((uint8_t*)device_frames + (size_t)i * frame_bytes, f->d_color, frame_bytes,
                                  hipMemcpyDeviceToDevice, f->cur_ms));
        if ((rc = finish_frame(ctx, f)) != WR_OK) return rc;
    }
    return WR_OK;
}

int wr_sync(WrContext* ctx) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    WR_HIP(hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != WR_OK) return rcs;
    }
    uint32_t e[4 * (WrContext::kMaxFrames + 1)] = {0};
    WR_HIP(hipMemcpy(e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost));
    ctx->stats.geometry_bin_items = e[4 * ctx->cur + 1];
    ctx->stats.shadow_bin_items = e[4 * ctx->cur + 2];
    if (const uint32_t es = e[4 * WrContext::kMaxFrames]) {
        WR_HIP(hipMemset(ctx->d_err + 4 * WrContext::kMaxFrames, 0, sizeof(uint32_t)));
        char msg[256];
        snprintf(msg, sizeof(msg), "%s capacity exceeded by a wr_render_frame_static call of this context (the latest was number %llu) since the last wr_sync",
                 (es & kErrClipOverflow) ? "clipped-triangle record" : "bin item", (unsigned long long)ctx->static_frames_submitted);
        return fail(ctx, WR_ERR_CAPACITY, msg);
    }
    // the earliest frame that overflowed a list
    int bad = -1;
    for (int k = 0; k < ctx->nfr; ++k)
        if (e[4 * k] && (bad < 0 || ctx->fr[k].serial < ctx->fr[bad].serial)) bad = k;
    if (bad >= 0) {
        for (int k = 0; k < ctx->nfr; ++k) WR_HIP(hipMemset(ctx->d_err + 4 * k, 0, sizeof(uint32_t)));
        char msg[256];
        snprintf(msg, sizeof(msg), "%s capacity exceeded by frame %llu of this context (of %llu submitted), or by a frame %d, %d, ... "
                                   "frames before it since the last wr_sync",
                 (e[4 * bad] & kErrClipOverflow) ? "clipped-triangle record" : "bin item",
                 (unsigned long long)ctx->fr[bad].serial, (unsigned long long)ctx->frames_submitted, ctx->nfr, 2 * ctx->nfr);
        return fail(ctx, WR_ERR_CAPACITY, msg);
    }
    return WR_OK;
}

static int read_back(WrContext* ctx, void* dst, size_t bytes, const void* src, size_t need) {
    if (!ctx || !dst) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null destination");
    if (bytes < need) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "destination too small");
    int rc = wr_sync(ctx);
    if (rc != WR_OK) return rc;
    WR_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    return WR_OK;
}
int wr_read_color(WrContext* ctx, uint8_t* dst, size_t bytes) {
    return read_back(ctx, dst, bytes, ctx ? ctx->fr[ctx->cur].d_color : nullptr, ctx ? (size_t)ctx->W * ctx->H * 4 : 0);
}
int wr_read_depth(WrContext* ctx, float* dst, size_t bytes) {
    return read_back(ctx, dst, bytes, ctx ? ctx->fr[ctx->cur].d_depth : nullptr, ctx ? (size_t)ctx->W * ctx->H * 4 : 0);
}
int wr_read_visibility(WrContext* ctx, uint32_t* dst, size_t bytes) {
    return read_back(ctx, dst, bytes, ctx ? ctx->fr[ctx->cur].d_vis : nullptr, ctx ? (size_t)ctx->W * ctx->H * 4 : 0);
}
int wr_read_shadow_map(WrContext* ctx, float* dst, size_t bytes) {
    return read_back(ctx, dst, bytes, ctx ? ctx->fr[ctx->cur].d_shadow : nullptr, ctx ? (size_t)6 * ctx->S * ctx->S * 4 : 0);
}

void* wr_color_device_ptr(WrContext* ctx) { return ctx ? ctx->fr[ctx->cur].d_color : nullptr; }

int wr_pack_owned_bins(WrContext* ctx, void* device_dst, uint32_t* out_bins) {
    if (!ctx || !device_dst) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null destination");
    WR_HIP(hipSetDevice(ctx->cfg.device));
    int bx = (ctx->W + kBin - 1) / kBin, by = (ctx->H + kBin - 1) / kBin, nb = bx * by;
    int world = (int)(ctx->cfg.tile_world ? ctx->cfg.tile_world : 1), rank = (int)ctx->cfg.tile_rank;
    int owned = (nb - rank + world - 1) / world;
    if (out_bins) *out_bins = (uint32_t)owned;
    FrameRes* f = &ctx->fr[ctx->cur];
    if (f->submitted) WR_HIP(hipStreamWaitEvent(ctx->stream, f->ev_done, 0));
    ctx->stream_has_work = true;
    if (owned > 0)
        hipLaunchKernelGGL(k_pack_bins, dim3((uint32_t)owned), dim3(256), 0, ctx->stream, f->d_color, (uint32_t*)device_dst,
                           ctx->W, ctx->H, bx, nb, rank, world, 1);
    WR_HIP(hipGetLastError());
    return WR_OK;
}

int wr_scatter_bins(WrContext* ctx, const void* device_src, uint32_t rank, uint32_t world) {
    if (!ctx || !device_src || world == 0 || rank >= world) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "bad scatter arguments");
    WR_HIP(hipSetDevice(ctx->cfg.device));
    int bx = (ctx->W + kBin - 1) / kBin, by = (ctx->H + kBin - 1) / kBin, nb = bx * by;
    int owned = (nb - (int)rank + (int)world - 1) / (int)world;
    FrameRes* f = &ctx->fr[ctx->cur];
    if (f->submitted) WR_HIP(hipStreamWaitEvent(ctx->stream, f->ev_done, 0));
    ctx->stream_has_work = true;
    if (owned > 0)
        hipLaunchKernelGGL(k_pack_bins, dim3((uint32_t)owned), dim3(256), 0, ctx->stream, f->d_color,
                           (uint32_t*)const_cast<void*>(device_src), ctx->W, ctx->H, bx, nb, (int)rank, (int)world, 0);
    WR_HIP(hipGetLastError());
    return WR_OK;
}

// ----------------------------------------------------------------------------- bin gather
void wr_gather_layout(uint32_t width, uint32_t height, uint32_t world, uint32_t rank, uint32_t* owned, uint32_t* slot_bins) {
    const uint32_t bx = (width + kBin - 1) / kBin, by = (height + kBin - 1) / kBin, nb = bx * by;
    if (world == 0) world = 1;
    if (owned) *owned = rank < world && nb > rank ? (nb - rank + world - 1) / world : 0;
    if (slot_bins) *slot_bins = (nb + world - 1) / world;
}

int wr_gather_bins_via(WrContext* ctx, const WrTransport* tp, uint32_t root) {
    if (!ctx || !tp || !tp->send || !tp->recv) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null transport");
    const uint32_t world = ctx->cfg.tile_world, rank = ctx->cfg.tile_rank;
    if (world < 2) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "the context is not bin-sharded (WrConfig.tile_world < 2)");
    if (root >= world) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "root rank out of range");
    WR_HIP(hipSetDevice(ctx->cfg.device));
    const int bx = (ctx->W + kBin - 1) / kBin, by = (ctx->H + kBin - 1) / kBin, nb = bx * by;
    const size_t bin_bytes = (size_t)kBin * kBin * 4;
    uint32_t owned = 0, slot = 0;
    wr_gather_layout((uint32_t)ctx->W, (uint32_t)ctx->H, world, rank, &owned, &slot);
    FrameRes* f = &ctx->fr[ctx->cur];
    if (f->submitted) WR_HIP(hipStreamWaitEvent(ctx->stream, f->ev_done, 0));
    ctx->stream_has_work = true;
    auto check = [&](int rc, const char* what) { return rc == 0 ? WR_OK : fail(ctx, WR_ERR_DEVICE, std::string("bin gather: ") + what + " failed"); };
    int rc;
    // Once a group is open it is closed whatever happens inside it (an open ncclGroupStart would
    // defer every later collective of the calling thread): the first failure is remembered,
    // group_end is called, then the failure is returned.
    if (rank != root) {
        if (owned)
            hipLaunchKernelGGL(k_pack_bins, dim3(owned), dim3(256), 0, ctx->stream, f->d_color, ctx->d_gather_send, ctx->W, ctx->H, bx, nb,
                               (int)rank, (int)world, 1);
        WR_HIP(hipGetLastError());
        if (tp->group_start && (rc = check(tp->group_start(tp->user), "group start")) != WR_OK) return rc;
        int first = WR_OK;
        if (owned) first = check(tp->send(tp->user, ctx->d_gather_send, owned * bin_bytes, (int)root, ctx->stream), "send");
        if (tp->group_end) {
            const std::string kept = ctx->err;
            rc = check(tp->group_end(tp->user), "group end");
            if (first != WR_OK) ctx->err = kept;
            else first = rc;
        }
        return first;
    }
    if (tp->group_start && (rc = check(tp->group_start(tp->user), "group start")) != WR_OK) return rc;
    {
        int first = WR_OK;
        for (uint32_t r = 0; r < world && first == WR_OK; ++r) {
            if (r == root) continue;
            uint32_t owned_r = 0;
            wr_gather_layout((uint32_t)ctx->W, (uint32_t)ctx->H, world, r, &owned_r, nullptr);
            if (!owned_r) continue;
            void* dst = reinterpret_cast<uint8_t*>(ctx->d_gather_recv) + (size_t)r * slot * bin_bytes;
            first = check(tp->recv(tp->user, dst, owned_r * bin_bytes, (int)r, ctx->stream), "receive");
        }
        if (tp->group_end) {
            const std::string kept = ctx->err;
            rc = check(tp->group_end(tp->user), "group end");
            if (first != WR_OK) ctx->err = kept;
            else first = rc;
        }
        if (first != WR_OK) return first;
    }
    hipLaunchKernelGGL(k_scatter_gathered, dim3((uint32_t)nb), dim3(256), 0, ctx->stream, f->d_color, ctx->d_gather_recv, ctx->W, ctx->H, bx,
                       nb, (int)world, (int)root, (int)slot);
    WR_HIP(hipGetLastError());
    return WR_OK;
}

// RCCL, loaded on first use: a host that never gathers does not need the library present.
extern "C++" {
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
}
constexpr int kNcclUint8 = 1;  // ncclUint8 of rccl.h
int rccl_group_start(void*) { return rccl().GroupStart(); }
int rccl_group_end(void*) { return rccl().GroupEnd(); }
int rccl_send(void* comm, const void* buf, size_t bytes, int peer, void* stream) {
    return rccl().Send(buf, bytes, kNcclUint8, peer, comm, (hipStream_t)stream);
}
int rccl_recv(void* comm, void* buf, size_t bytes, int peer, void* stream) {
    return rccl().Recv(buf, bytes, kNcclUint8, peer, comm, (hipStream_t)stream);
}
}  // namespace
}  // extern "C++"

int wr_gather_bins(WrContext* ctx, void* nccl_comm, uint32_t root) {
    if (!ctx || !nccl_comm) return fail(ctx, WR_ERR_INVALID_ARGUMENT, "null communicator");
    if (!rccl().ok) return fail(ctx, WR_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    WrTransport tp;
    const int rc = wr_comm_transport(nccl_comm, &tp);
    if (rc != WR_OK) return fail(ctx, rc, "no RCCL transport");
    return wr_gather_bins_via(ctx, &tp, root);
}
int wr_comm_unique_id(uint8_t id[128]) {
    if (!id) return WR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return WR_ERR_UNSUPPORTED;
    return rccl().GetUniqueId(id) == 0 ? WR_OK : WR_ERR_DEVICE;
}
int wr_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, int32_t device, void** out_comm) {
    if (!id || !out_comm || rank >= world) return WR_ERR_INVALID_ARGUMENT;
    *out_comm = nullptr;
    if (!rccl().ok) return WR_ERR_UNSUPPORTED;
    if (hipSetDevice(device) != hipSuccess) return WR_ERR_DEVICE;
    ncclUniqueIdBytes uid;
    memcpy(uid.internal, id, sizeof(uid.internal));
    return rccl().CommInitRank(out_comm, (int)world, uid, (int)rank) == 0 ? WR_OK : WR_ERR_DEVICE;
}
int wr_comm_destroy(void* comm) {
    if (!comm) return WR_OK;
    if (!rccl().ok) return WR_ERR_UNSUPPORTED;
    return rccl().CommDestroy(comm) == 0 ? WR_OK : WR_ERR_DEVICE;
}
int wr_comm_transport(void* comm, WrTransport* out) {
    if (!comm || !out) return WR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return WR_ERR_UNSUPPORTED;
    *out = WrTransport{comm, rccl_group_start, rccl_send, rccl_recv, rccl_group_end};
    return WR_OK;
}
int wr_comm_library_path(char* path, size_t cap) {
    if (!path || cap == 0) return WR_ERR_INVALID_ARGUMENT;
    path[0] = 0;
    if (!rccl().ok) return WR_ERR_UNSUPPORTED;
    Dl_info info;
    memset(&info, 0, sizeof(info));
    if (!dladdr(reinterpret_cast<void*>(rccl().Send), &info) || !info.dli_fname) return WR_ERR_DEVICE;
    snprintf(path, cap, "%s", info.dli_fname);
    return WR_OK;
}

int wr_set_profiling(WrContext* ctx, int enabled) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    ctx->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    ctx->ev_level = ctx->profiling;
    ctx->ev_count = 0;
    ctx->ev_next = 0;
    ctx->prof_tick = 0;
    return WR_OK;
}
int wr_set_counting(WrContext* ctx, int enabled) {
    if (!ctx) return WR_ERR_INVALID_ARGUMENT;
    ctx->counting = enabled != 0;
    return WR_OK;
}

int wr_get_stats(WrContext* ctx, WrFrameStats* out) {
    if (!ctx || !out) return WR_ERR_INVALID_ARGUMENT;
    int rc = wr_sync(ctx);
    if (rc != WR_OK) return rc;
    if (ctx->ev_count > 0) {
        // shadow runs next to geometry and fog next to shade (side stream); shade includes any
        // wait for the shadow maps, the frame ends with the fog blend
        const int a[7] = {1, 0, 3, 8, 0, 6, 10}, b[7] = {2, 3, 4, 5, 7, 9, 5};  // shadow, geometry, shade, fog, frame, view noise, light noise
        double ms[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < ctx->ev_count; ++k)
            for (int i = (ctx->ev_level == 2 ? 5 : 0); i < (ctx->ev_level == 2 ? 6 : 7); ++i) {  // (level 2 recorded the view noise / march pair only)
                float t = 0.0f;
                WR_HIP(hipEventElapsedTime(&t, ctx->ev[k][a[i]], ctx->ev[k][b[i]]));
                ms[i] += t;
            }
        for (int i = 0; i < 7; ++i) ms[i] /= ctx->ev_count;  // (zero where nothing was recorded)
        ctx->stats.ms_shadow = (float)ms[0]; ctx->stats.ms_geometry = (float)ms[1];
        ctx->stats.ms_shade = (float)ms[2]; ctx->stats.ms_fog = (float)ms[3];
        ctx->stats.ms_total = (float)ms[4];
        ctx->stats.ms_fog_march = (float)ms[5];
        ctx->stats.ms_fog_light_march = (float)ms[6];
        ctx->stats.profiled_frames = (uint32_t)ctx->ev_count;
    }
    unsigned long long c[8];
    WR_HIP(hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    ctx->stats.fog_fragments = c[0]; ctx->stats.fog_noise_evals = c[1];
    ctx->stats.shaded_fragments = c[2]; ctx->stats.replay_pixels = c[3]; ctx->stats.replay_slow_pixels = c[4];
    ctx->stats.fog_noise_evals_view = c[5]; ctx->stats.fog_noise_evals_light = c[6];
    *out = ctx->stats;
    return WR_OK;
}

// ----------------------------------------------------------------------------- host scene math

static void look_to_rh(const float eye[3], const float dir[3], const float up[3], float m[4][4]) {
    // cgmath Matrix4::look_to_rh: f = normalize(dir), s = normalize(f x up), u = s x f
    float fl = 1.0f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    float f[3] = {dir[0] * fl, dir[1] * fl, dir[2] * fl};
    float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    float sl = 1.0f / sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    s[0] *= sl; s[1] *= sl; s[2] *= sl;
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    m[0][0] = s[0]; m[0][1] = u[0]; m[0][2] = -f[0]; m[0][3] = 0.0f;
    m[1][0] = s[1]; m[1][1] = u[1]; m[1][2] = -f[1]; m[1][3] = 0.0f;
    m[2][0] = s[2]; m[2][1] = u[2]; m[2][2] = -f[2]; m[2][3] = 0.0f;
    m[3][0] = -(eye[0] * s[0] + eye[1] * s[1] + eye[2] * s[2]);
    m[3][1] = -(eye[0] * u[0] + eye[1] * u[1] + eye[2] * u[2]);
    m[3][2] = eye[0] * f[0] + eye[1] * f[1] + eye[2] * f[2];
    m[3][3] = 1.0f;
}

static void perspective(float fovy_deg, float aspect, float zn, float zf, float m[4][4]) {
    // cgmath::perspective (GL-style clip z in [-w, w]); Deg -> Rad multiplies by f32(pi/180)
    float rad = fovy_deg * (float)(3.14159265358979323846 / 180.0);
    float f = 1.0f / tanf(rad / 2.0f);
    memset(m, 0, 64);
    m[0][0] = f / aspect;
    m[1][1] = f;
    m[2][2] = (zf + zn) / (zn - zf);
    m[2][3] = -1.0f;
    m[3][2] = (2.0f * zf * zn) / (zn - zf);
}

static bool invert4(const float* a, float* out) {
    double m[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { m[r][c] = a[4 * c + r]; m[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int p = i;
        for (int r = i + 1; r < 4; ++r) if (fabs(m[r][i]) > fabs(m[p][i])) p = r;
        if (m[p][i] == 0.0) return false;
        for (int c = 0; c < 8; ++c) std::swap(m[i][c], m[p][c]);
        double inv = 1.0 / m[i][i];
        for (int c = 0; c < 8; ++c) m[i][c] *= inv;
        for (int r = 0; r < 4; ++r) if (r != i) { double f = m[r][i]; for (int c = 0; c < 8; ++c) m[r][c] -= f * m[i][c]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * c + r] = (float)m[r][4 + c];
    return true;
}

static const float kNear = 1.0f, kFar = 3000.0f;  // camera.rs:8-9

void wr_camera_uniform_update(WrCameraUniform* out, const float position[3], float pitch_deg, float yaw_deg,
                              float fovy_deg, uint32_t width, uint32_t height) {
    const float d2r = (float)(3.14159265358979323846 / 180.0);
    float ys = sinf(yaw_deg * d2r), yc = cosf(yaw_deg * d2r);
    float ps = sinf(pitch_deg * d2r), pc = cosf(pitch_deg * d2r);
    float fwd[3] = {pc * yc, ps, pc * ys};
    float fl = 1.0f / sqrtf(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]);
    fwd[0] *= fl; fwd[1] *= fl; fwd[2] *= fl;
    float right[3] = {-ys, 0.0f, yc};
    float rl = 1.0f / sqrtf(right[0] * right[0] + right[1] * right[1] + right[2] * right[2]);
    right[0] *= rl; right[1] *= rl; right[2] *= rl;
    float up[3] = {right[1] * fwd[2] - right[2] * fwd[1], right[2] * fwd[0] - right[0] * fwd[2], right[0] * fwd[1] - right[1] * fwd[0]};
    look_to_rh(position, fwd, up, out->view);
    perspective(fovy_deg, (float)width / (float)height, kNear, kFar, out->proj);
    float pv[16];
    mat4_mul_mat4(&out->proj[0][0], &out->view[0][0], pv);
    if (!invert4(pv, &out->inv_view_proj[0][0])) memset(out->inv_view_proj, 0, 64);
    out->position[0] = position[0]; out->position[1] = position[1]; out->position[2] = position[2]; out->position[3] = 1.0f;
    out->planes[0] = kNear; out->planes[1] = kFar; out->planes[2] = (float)width; out->planes[3] = (float)height;
}

void wr_light_update_matrices(WrLightUniform* light) {
    float proj[4][4];
    perspective(90.0f, 1.0f, kNear, kFar, proj);
    static const float dirs[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    static const float ups[6][3] = {{0, 1, 0}, {0, 1, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, 1, 0}};
    for (int i = 0; i < 6; ++i) {
        float view[4][4];
        look_to_rh(light->position, dirs[i], ups[i], view);
        mat4_mul_mat4(&proj[0][0], &view[0][0], &light->matrices[i][0][0]);
    }
}

void wr_light_animate(WrLightUniform* light, float t) {
    light->position[0] = sinf(t * 0.5f) * 500.0f;
    light->position[1] = 250.0f + sinf(t * 0.3f) * 200.0f;
    light->position[2] = sinf(t * 0.8f) * 100.0f;
    wr_light_update_matrices(light);
    light->color[0] = fabsf(sinf(t * 1.0f));
    light->color[1] = fabsf(sinf(t * 0.6f));
    light->color[2] = fabsf(sinf(t * 0.4f));
}

void wr_instance_to_raw(WrInstanceRaw* out, const float position[3], const float q[4], const float scale[3]) {
    // cgmath Matrix3::from(Quaternion{s, v})
    float s = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
    float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
    float r[3][3] = {{1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2}, {xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2}, {xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2}};
    for (int c = 0; c < 3; ++c) {
        for (int rr = 0; rr < 3; ++rr) { out->normal[c][rr] = r[c][rr]; out->model[c][rr] = scale[rr] * r[c][rr]; }
        out->model[c][3] = 0.0f;
    }
    out->model[3][0] = position[0]; out->model[3][1] = position[1]; out->model[3][2] = position[2]; out->model[3][3] = 1.0f;
}

void wr_camera_fly_update(float pose[5], const float in[5], float speed, float sensitivity, float dt) {
    float pitch = pose[3] - in[1] * sensitivity * 0.022f;
    pitch = pitch < -89.0f ? -89.0f : (pitch > 89.0f ? 89.0f : pitch);
    float yaw = pose[4] + in[0] * sensitivity * 0.022f;
    yaw = fmodf(yaw, 360.0f);
    if (yaw < 0.0f) yaw += 360.0f;
    const float d2r = (float)(3.14159265358979323846 / 180.0);
    float ys = sinf(yaw * d2r), yc = cosf(yaw * d2r), ps = sinf(pitch * d2r), pc = cosf(pitch * d2r);
    float fwd[3] = {pc * yc, ps, pc * ys};
    float fl = 1.0f / sqrtf(fwd[0] * fwd[0] + fwd[1] * fwd[1] + fwd[2] * fwd[2]);
    fwd[0] *= fl; fwd[1] *= fl; fwd[2] *= fl;
    float right[3] = {-ys, 0.0f, yc};
    float up[3] = {right[1] * fwd[2] - right[2] * fwd[1], right[2] * fwd[0] - right[0] * fwd[2], right[0] * fwd[1] - right[1] * fwd[0]};
    for (int k = 0; k < 3; ++k) {
        pose[k] += fwd[k] * in[2] * speed * dt;
        pose[k] += right[k] * in[3] * speed * dt;
        pose[k] += up[k] * in[4] * speed * dt;
    }
    pose[3] = pitch; pose[4] = yaw;
}

}  // extern "C"

#ifdef WR_DEV
// Developer build only: copies binning state of the last frame to the host.
extern "C" int wr_dev_read(WrContext* ctx, int which, void* dst, size_t bytes) {
    FrameRes* f = &ctx->fr[ctx->cur];
    WsAlloc& w = (which & 1) ? f->geom : f->shadow;
    const RasterWS& ws = w.ws;
    size_t nb = (size_t)ws.bins_x * ws.bins_y * ws.layers;
    size_t nrec = (size_t)ws.layers * ((size_t)ws.slots + ws.ovf_cap);
    const void* src = nullptr;
    size_t have = 0;
    switch (which >> 1) {
        case 0: src = ws.bin_count; have = nb * 4; break;
        case 1: src = ws.bb; have = nrec * 8; break;
        case 2: src = ws.bin_offset; have = nb * 4; break;
        case 3: src = ws.bin_items; have = (size_t)ws.items_cap * 4; break;
    }
    (void)sync_all(ctx);
    if ((which >> 1) == 5) {
        have = sizeof(unsigned long long) * 2 * 16 * 2;
        if (bytes > have) bytes = have;
        if (bytes && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg_levels), bytes) != hipSuccess) return -1;
        return (int)(have / 4);
    }
    if ((which >> 1) == 8) {
        have = sizeof(unsigned long long) * 8;
        if (bytes > have) bytes = have;
        if (bytes && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg_blocks), bytes) != hipSuccess) return -1;
        return (int)(have / 4);
    }
    if ((which >> 1) == 7) {
        have = sizeof(unsigned long long) * 3 * 8;
        if (bytes > have) bytes = have;
        if (bytes && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg_phase), bytes) != hipSuccess) return -1;
        return (int)(have / 4);
    }
    if ((which >> 1) == 6) {
        have = sizeof(unsigned long long) * 2 * 16 * 6;
        if (bytes > have) bytes = have;
        if (bytes && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg_boxes), bytes) != hipSuccess) return -1;
        return (int)(have / 4);
    }
    if ((which >> 1) == 4) {
        have = sizeof(unsigned long long) * 3 * 40000;
        if (bytes > have) bytes = have;
        if (bytes && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg_cycles), bytes) != hipSuccess) return -1;
        return (int)(have / 4);
    }
    if (bytes > have) bytes = have;
    if (hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)(have / 4 > 0x7fffffff ? 0x7fffffff : have / 4);
}
#endif
