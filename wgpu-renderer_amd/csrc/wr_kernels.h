// wr_kernels.h -- gfx950 kernels of the render path.
//
// Frame structure (kernels of one frame; the host runs the shadow pass and the fog pass up to
// the march on a side stream, see frame_stages in wr_api.hip):
//   shadow   : k_setup<true> (vertex stage + set-up + bin counts) -> k_setup_clip<true> ->
//              k_scan -> k_bin_fill -> k_raster<kRasterShadow>   (6 cube faces as grid.y "layers")
//   geometry : k_vs_geometry -> k_setup<false> -> k_setup_clip<false> (+ k_setup_small for the
//              light marker) -> k_scan -> k_bin_fill -> k_raster<kRasterColor>
//              (visibility + depth buffers)
//   shade    : k_shade -> k_raster<kRasterLayers> -> k_resolve_shade -> k_replay_slow
//              (deferred pbr.wgsl fragment stage; ordered blending of translucent fronts)
//   fog      : k_setup_small(fog) -> k_fog_tables -> k_fog_setup -> k_fog_noise_view ->
//              k_fog_view_resolve -> k_fog_noise_light -> k_fog_shade_blend (+ k_fog_multi)
//                                                                 (fog.wgsl fragment stage)
//
// The rasteriser is tile-binned: the target is cut into 64x64-pixel bins, every set-up
// triangle is appended to the list of each bin it can touch, and one workgroup per bin
// resolves visibility in an LDS tile (64-bit keys: depth bits << 32 | triangle id, combined
// with LDS atomic-min), then writes the tile out once with coalesced stores.  Small triangles
// are rasterised one per lane; a large one is taken by a whole wavefront, whose 64 lanes test
// one 8x8 pixel block per step against a wave-uniform record held in scalar registers.
#pragma once

#include "wr_device.h"

namespace wr {

// Issue priority of a wavefront among those resident on its SIMD.  Every kernel but the two
// noise kernels of the fog march is short and waits on memory or atomics most of the time; the
// noise kernels keep every wave slot of the chip issuing VALU instructions for most of a frame.
// When they share a SIMD, a round-robin issue order makes each instruction of the short kernel
// wait for all the marching waves' turns, which stretches it several times over while the march
// gains nothing.  With the raised priority the short kernels issue whenever they are ready and
// the march fills the cycles in between.
#ifndef WR_LATENCY_PRIORITY
#define WR_LATENCY_PRIORITY 3
#endif
__device__ inline void latency_bound_kernel() { __builtin_amdgcn_s_setprio(WR_LATENCY_PRIORITY); }

struct RasterWS {
    Rec* recs;            // [layers][slots + ovf_cap]
    uint2* bb;            // [layers][slots + ovf_cap] pixel bounding box of each record, or (kBBInvalid, kBB*)
    float* betas;         // [layers][ovf_cap][9] clip weights of records born in the clipper
    uint32_t* ovf_count;  // [layers]
    uint32_t* bin_count;  // [layers * bins]
    uint32_t* bin_cursor; // [layers * bins]
    uint32_t* bin_offset; // [layers * bins]
    uint32_t* bin_items;  // record indices, grouped by bin
    uint32_t* total_items;
    uint32_t* scan_total;   // k_scan: entries of the item list handed out so far this frame
    uint32_t* err;
    uint32_t* clip_count; // [layers] triangles queued for the clipper this frame
    uint32_t* clip_queue; // [layers][slots] their triangle indices
    uint32_t slots, ovf_cap, items_cap;
    int bins_x, bins_y, layers;
};

enum : uint32_t { kErrClipOverflow = 1u, kErrItemOverflow = 2u };

// ------------------------------------------------------------------ vertex stages
struct VertexIn {
    const float4* pos;   // xyz1
    const float2* uv;
    const float4* nrm;
    const float4* tan;
    const float4* bit;
    uint32_t nv;
};
struct VertexOut {       // VertexOutput of globals.wgsl:55-62, packed in four float4 streams
    float4* clip;
    float4* v0;          // uv.xy, tpos.xy
    float4* v1;          // tpos.z, tlight.xyz
    float4* v2;          // tview.xyz, wpos.x
    float4* v3;          // wpos.yzw, 0
};

// pbr.wgsl:8-46
__global__ __launch_bounds__(256) void k_vs_geometry(VertexIn in, VertexOut out, FrameConsts fc) {
    latency_bound_kernel();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.nv) return;
    float4 p = in.pos[i], n = in.nrm[i], t = in.tan[i], b = in.bit[i];
    float2 uv = in.uv[i];
    V3 wn = normalize3(mat3_mul(fc.normal_geom, v3(n.x, n.y, n.z)));
    V3 wt = normalize3(mat3_mul(fc.normal_geom, v3(t.x, t.y, t.z)));
    V3 wb = normalize3(mat3_mul(fc.normal_geom, v3(b.x, b.y, b.z)));
    V4 wp = mat4_mul(fc.model_geom, V4{p.x, p.y, p.z, 1.0f});
    V4 cl = mat4_mul(fc.pv, wp);
    V3 w3 = v3(wp.x, wp.y, wp.z);
    V3 lp = v3(fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]);
    V3 cp = v3(fc.cam_pos[0], fc.cam_pos[1], fc.cam_pos[2]);
    out.clip[i] = make_float4(cl.x, cl.y, cl.z, cl.w);
    out.v0[i] = make_float4(uv.x, uv.y, dot3(wt, w3), dot3(wb, w3));
    out.v1[i] = make_float4(dot3(wn, w3), dot3(wt, lp), dot3(wb, lp), dot3(wn, lp));
    out.v2[i] = make_float4(dot3(wt, cp), dot3(wb, cp), dot3(wn, cp), wp.x);
    out.v3[i] = make_float4(wp.y, wp.z, wp.w, 0.0f);
}

// ------------------------------------------------------------------ triangle set-up
__device__ inline void store_rec(Rec* dst, const Rec& r) {
    const uint4* s = reinterpret_cast<const uint4*>(&r);
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}
__device__ inline Rec load_rec(const Rec* src) {
    Rec r;
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(&r);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
    return r;
}

// bb.x == kBBInvalid: the slot holds no raster record; bb.y then says whether its Rec is the
// parent of clipped fan triangles (first index, count) or nothing at all.  A slot without a
// record is marked here only: its 64-byte Rec is not touched (a 4-byte store into it costs a
// whole line of HBM write traffic, six faces x every culled triangle per frame).
constexpr uint32_t kBBInvalid = 0xFFFFFFFFu;
constexpr uint32_t kBBNothing = 0u, kBBClippedParent = 1u;

// Adds 1 to counters[key] for every lane with `active`, with one atomic per distinct key in
// the wavefront (ballot + leader), and returns to each such lane the value its own
// atomicAdd(…, 1) would have returned in lane order.  Neighbouring triangles of a mesh land
// in the same bin, so this removes almost all same-address atomic traffic.  Every lane that
// has not exited the kernel must call it.
__device__ inline uint32_t wave_agg_inc(uint32_t* counters, bool active, uint32_t key) {
    const int lane = (int)(threadIdx.x & 63u);
    uint32_t pos = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t lkey = (uint32_t)__shfl((int)key, leader);
        const bool mine = active && key == lkey;
        const unsigned long long m = __ballot(mine);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counters[lkey], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (mine) pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return pos;
}

// Appends `want` entries per thread to a global list: returns the index of the thread's first
// entry.  One global atomic per workgroup (atomics of a whole grid on one address are
// serialised at the memory side and cost far more than anything else these kernels do), an
// LDS atomic per wavefront.  s_cnt / s_base are two LDS words of the caller; every thread of
// the workgroup must call this (two barriers inside).
__device__ inline uint32_t group_append(uint32_t* counter, uint32_t want, uint32_t* s_cnt, uint32_t* s_base) {
    const int lane = (int)(threadIdx.x & 63u);
    if (threadIdx.x == 0) *s_cnt = 0u;
    __syncthreads();
    uint32_t incl = want;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += v;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    uint32_t wbase = 0;
    if (lane == 0 && total) wbase = atomicAdd(s_cnt, total);
    wbase = (uint32_t)__shfl((int)wbase, 0);
    __syncthreads();
    if (threadIdx.x == 0) *s_base = *s_cnt ? atomicAdd(counter, *s_cnt) : 0u;
    __syncthreads();
    return *s_base + wbase + incl - want;
}

__device__ inline bool bin_owned(int b, int tile_rank, int tile_world) {
    return tile_world <= 1 || (b % tile_world) == tile_rank;
}

// Bins of a record whose bounding box spans several bins: every bin of the box that the
// triangle's edges do not rule out.  FILL = false counts, FILL = true appends the record
// index; both make the same decisions.
template <bool FILL>
__device__ inline void bin_record_multi(const RasterWS& ws, uint32_t layer, uint32_t rec_index, const Rec& r,
                                        int tile_rank, int tile_world) {
    int pxmin = (int)(r.bbmin & 0xffffu), pymin = (int)(r.bbmin >> 16);
    int pxmax = (int)(r.bbmax & 0xffffu), pymax = (int)(r.bbmax >> 16);
    Edges e;
    make_edges(r, e);
    const uint32_t nb = (uint32_t)(ws.bins_x * ws.bins_y);
    for (int by = pymin / kBin; by <= pymax / kBin; ++by)
        for (int bx = pxmin / kBin; bx <= pxmax / kBin; ++bx) {
            int b = by * ws.bins_x + bx;
            if (!bin_owned(b, tile_rank, tile_world)) continue;
            int x0 = imax(bx * kBin, pxmin), x1 = imin(bx * kBin + kBin - 1, pxmax);
            int y0 = imax(by * kBin, pymin), y1 = imin(by * kBin + kBin - 1, pymax);
            if (box_rejected(e, x0, y0, x1, y1)) continue;
            uint32_t gb = layer * nb + (uint32_t)b;
            if (!FILL) {
                atomicAdd(&ws.bin_count[gb], 1u);
            } else {
                uint32_t o = ws.bin_offset[gb] + atomicAdd(&ws.bin_cursor[gb], 1u);
                if (o < ws.items_cap) ws.bin_items[o] = rec_index;
            }
        }
}

__device__ inline bool bb_single_bin(uint32_t bbmin, uint32_t bbmax, int bins_x, int& b) {
    int bx0 = (int)(bbmin & 0xffffu) / kBin, by0 = (int)(bbmin >> 16) / kBin;
    int bx1 = (int)(bbmax & 0xffffu) / kBin, by1 = (int)(bbmax >> 16) / kBin;
    b = by0 * bins_x + bx0;
    return bx0 == bx1 && by0 == by1;
}

struct SetupTarget {
    Rec* recs;          // this layer's records
    uint2* bb;          // this layer's bounding boxes (bbmin = kBBInvalid: nothing to bin)
    float* betas;
    uint32_t* ovf_count;
    uint32_t layer;
    int W, H, tile_rank, tile_world;
    bool count;         // false: no binning for this target (the fog volume is not binned)
};

// Clip path: the fan triangles go to the overflow region of the record array, the parent
// slot keeps where they are.  Rare, so its bins are counted without aggregation.
// `work` (optional): room for twenty vertices the clipper may use instead of per-lane arrays.
template <bool SHADOW>
__device__ __attribute__((noinline)) void emit_clipped(const V4* c, uint32_t tri, uint32_t slot, const RasterWS& ws,
                                                       const SetupTarget& tg, ClipVert* work = nullptr) {
    ClipVert own[20];
    ClipVert* poly = work ? work : own;
    for (int k = 0; k < 3; ++k) {
        poly[k].x = c[k].x; poly[k].y = c[k].y; poly[k].z = c[k].z; poly[k].w = c[k].w;
        poly[k].b0 = k == 0 ? 1.0f : 0.0f; poly[k].b1 = k == 1 ? 1.0f : 0.0f; poly[k].b2 = k == 2 ? 1.0f : 0.0f;
    }
    int n = clip_polygon_in(poly, poly + 10, 3);
    Rec parent;
    parent.id = kRecInvalid;
    if (n >= 3) {
        uint32_t nf = (uint32_t)(n - 2);
        uint32_t base = atomicAdd(tg.ovf_count, nf);
        if (base + nf > ws.ovf_cap) {
            atomicOr(ws.err, kErrClipOverflow);
        } else {
            for (uint32_t i = 0; i < nf; ++i) {
                ClipVert cv[3] = {poly[0], poly[i + 1], poly[i + 2]};
                Rec r;
                float beta[9];
                uint32_t ri = ws.slots + base + i;
                if (setup_sub(cv, tg.W, tg.H, SHADOW, (tri << 3) | i, r, beta)) {
                    store_rec(&tg.recs[ri], r);
                    tg.bb[ri] = make_uint2(r.bbmin, r.bbmax);
                    float* bd = tg.betas + (size_t)(base + i) * 9;
                    for (int k = 0; k < 9; ++k) bd[k] = beta[k];
                    if (tg.count) {
                        // same rule as k_bin_fill: a box inside one bin is binned without the edge test
                        int b = 0;
                        if (bb_single_bin(r.bbmin, r.bbmax, ws.bins_x, b)) {
                            if (bin_owned(b, tg.tile_rank, tg.tile_world))
                                atomicAdd(&ws.bin_count[tg.layer * (uint32_t)(ws.bins_x * ws.bins_y) + (uint32_t)b], 1u);
                        } else {
                            bin_record_multi<false>(ws, tg.layer, ri, r, tg.tile_rank, tg.tile_world);
                        }
                    }
                } else {
                    tg.bb[ri] = make_uint2(kBBInvalid, kBBNothing);
                }
            }
            parent.id = kRecClippedParent;
            parent.X0 = (int32_t)(ws.slots + base);
            parent.Y0 = (int32_t)nf;
        }
    }
    if (parent.id == kRecClippedParent) store_rec(&tg.recs[slot], parent);
    tg.bb[slot] = make_uint2(kBBInvalid, parent.id == kRecClippedParent ? kBBClippedParent : kBBNothing);
}

// Sets up one triangle, stores its record (only if something can be covered), its bounding
// box, and counts it into the bins it touches.  `have` is false for lanes past the end of
// the triangle list; they still take part in the wave-aggregated counting.
// With DEFER_CLIP a triangle that needs the clipper is left alone and reported to the caller
// (return value), which queues it for k_setup_clip: the clipper's arrays and registers then
// do not weigh on the occupancy of the kernel that sets up the other 99 %.
template <bool SHADOW, bool DEFER_CLIP>
__device__ inline bool emit_triangle(bool have, const V4* c, uint32_t tri, uint32_t slot, const RasterWS& ws,
                                     const SetupTarget& tg, ClipVert* clip_work = nullptr) {
    Rec r;
    r.id = kRecInvalid;
    bool valid = false, deferred = false;
    if (have) {
        int cls = classify_triangle(c);
        if (cls == 2) {
            if (DEFER_CLIP) deferred = true;
            else emit_clipped<SHADOW>(c, tri, slot, ws, tg, clip_work);
        } else {
            if (cls == 1) {
                ClipVert cv[3];
                for (int k = 0; k < 3; ++k) { cv[k].x = c[k].x; cv[k].y = c[k].y; cv[k].z = c[k].z; cv[k].w = c[k].w; cv[k].b0 = cv[k].b1 = cv[k].b2 = 0.0f; }
                valid = setup_sub(cv, tg.W, tg.H, SHADOW, tri << 3, r, nullptr);
            }
            if (valid) {
                store_rec(&tg.recs[slot], r);
                tg.bb[slot] = make_uint2(r.bbmin, r.bbmax);
            } else {
                tg.bb[slot] = make_uint2(kBBInvalid, kBBNothing);
            }
        }
    }
    if (!tg.count) return deferred;
    int b = 0;
    const bool single = valid && bb_single_bin(r.bbmin, r.bbmax, ws.bins_x, b);
    const uint32_t gb = tg.layer * (uint32_t)(ws.bins_x * ws.bins_y) + (uint32_t)b;
    wave_agg_inc(ws.bin_count, single && bin_owned(b, tg.tile_rank, tg.tile_world), gb);
    if (valid && !single) bin_record_multi<false>(ws, tg.layer, slot, r, tg.tile_rank, tg.tile_world);
    return deferred;
}

__device__ inline SetupTarget setup_target(const RasterWS& ws, uint32_t layer, int W, int H, int tile_rank, int tile_world) {
    SetupTarget tg;
    size_t stride = (size_t)ws.slots + ws.ovf_cap;
    tg.recs = ws.recs + layer * stride;
    tg.bb = ws.bb + layer * stride;
    tg.betas = ws.betas + (size_t)layer * ws.ovf_cap * 9;
    tg.ovf_count = ws.ovf_count + layer;
    tg.layer = layer;
    tg.W = W; tg.H = H; tg.tile_rank = tile_rank; tg.tile_world = tile_world;
    tg.count = true;
    return tg;
}

// The shadow passes' vertex stage (depth.wgsl:3-17) is done by the triangle set-up itself: a
// triangle transforms its three model-space positions with its face's light matrix.  The same
// two products, in the same order, as a vertex kernel would do once per vertex and face: the
// extra multiplications (a vertex is used by six triangles) cost less than writing and reading
// 16 bytes per vertex and face, and the frame's chain of launches is one shorter.
struct ShadowXf {
    float model[16];     // InstanceRaw.model of the geometry model
    float light[6][16];  // LightUniform.matrices
};
// Clip-space position of vertex i of layer `layer`: the stored one of the colour pass (`v` =
// clip positions), or the transformed model-space position (`v` = positions) of a shadow face.
template <bool SHADOW>
__device__ inline V4 setup_vertex(const float4* v, uint32_t i, uint32_t layer, const ShadowXf& xf) {
    const float4 p = v[i];
    if (!SHADOW) return V4{p.x, p.y, p.z, p.w};
    const V4 wp = mat4_mul(xf.model, V4{p.x, p.y, p.z, 1.0f});
    return mat4_mul(xf.light[layer], wp);
}

// One thread per triangle of the geometry model; grid.y = layer (cube face for shadows).
template <bool SHADOW>
__global__ __launch_bounds__(256) void k_setup(const float4* verts, const uint32_t* idx, uint32_t ntris,
                                               RasterWS ws, int W, int H, int tile_rank, int tile_world, ShadowXf xf) {
    latency_bound_kernel();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = t < ntris;
    uint32_t layer = blockIdx.y;
    V4 c[3];
    if (have)
        for (int k = 0; k < 3; ++k) c[k] = setup_vertex<SHADOW>(verts, idx[3 * t + k], layer, xf);
    __shared__ uint32_t s_cnt, s_base;
    SetupTarget tg = setup_target(ws, layer, W, H, tile_rank, tile_world);
    const bool deferred = emit_triangle<SHADOW, true>(have, c, t, t, ws, tg);
    if (!__syncthreads_or(deferred ? 1 : 0)) return;
    const uint32_t k = group_append(&ws.clip_count[layer], deferred ? 1u : 0u, &s_cnt, &s_base);
    if (deferred) ws.clip_queue[(size_t)layer * ws.slots + k] = t;
}

// The shadow passes' set-up in two steps.  A triangle lies in the frustum of one or two of the six
// cube faces and outside the others, so a thread per (triangle, face) pair fetched every triangle
// six times to drop it five times.  k_shadow_classify fetches a triangle once, classifies it
// against all six faces (the same two matrix products per vertex and face and the same
// classify_triangle as the set-up proper, so the same decisions), marks the slots of the faces
// that drop it and lists the other (triangle, face) pairs; k_setup_pairs sets those up, one
// thread per listed pair.  The list's length is only known on the device: the grid has a fixed
// size and its workgroups stride over the list.
constexpr uint32_t kPairFaceShift = 29;  // pair = triangle | face << 29 (a context holds fewer than 2^29 triangles)
__global__ __launch_bounds__(256) void k_shadow_classify(const float4* verts, const uint32_t* idx, uint32_t ntris, RasterWS ws,
                                                         ShadowXf xf, uint32_t* pair_count, uint32_t* pairs) {
    latency_bound_kernel();
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t faces = 0u;
    if (t < ntris) {
        V4 wp[3];
        for (int k = 0; k < 3; ++k) {
            const float4 p = verts[idx[3 * t + k]];
            wp[k] = mat4_mul(xf.model, V4{p.x, p.y, p.z, 1.0f});
        }
        const size_t stride = (size_t)ws.slots + ws.ovf_cap;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            V4 c[3];
            for (int k = 0; k < 3; ++k) c[k] = mat4_mul(xf.light[f], wp[k]);
            if (classify_triangle(c) != 0) faces |= 1u << f;
            else ws.bb[(size_t)f * stride + t] = make_uint2(kBBInvalid, kBBNothing);
        }
    }
    uint32_t k = group_append(pair_count, (uint32_t)__builtin_popcount(faces), &s_cnt, &s_base);
#pragma unroll
    for (uint32_t f = 0; f < 6u; ++f)
        if ((faces >> f) & 1u) pairs[k++] = t | (f << kPairFaceShift);
}
__global__ __launch_bounds__(256) void k_setup_pairs(const float4* verts, const uint32_t* idx, RasterWS ws, int W, int H, ShadowXf xf,
                                                     const uint32_t* pair_count, const uint32_t* pairs) {
    latency_bound_kernel();
    const uint32_t n = *pair_count;
    // (the grid is sized for about two faces per triangle; a workgroup takes every gridDim.x-th
    // group of 256 pairs, all its threads going round together)
    for (uint32_t k0 = blockIdx.x * blockDim.x; k0 < n; k0 += gridDim.x * blockDim.x) {
        const uint32_t k = k0 + threadIdx.x;
        const bool have = k < n;
        const uint32_t pr = have ? pairs[k] : 0u;
        const uint32_t t = pr & ((1u << kPairFaceShift) - 1u), layer = pr >> kPairFaceShift;
        V4 c[3];
        if (have)
            for (int j = 0; j < 3; ++j) c[j] = setup_vertex<true>(verts, idx[3 * t + j], layer, xf);
        SetupTarget tg = setup_target(ws, layer, W, H, 0, 1);
        const bool deferred = emit_triangle<true, true>(have, c, t, t, ws, tg);
        // triangles that need the clipper: queued per face, one atomic per face and wavefront
        if (__any(deferred ? 1 : 0)) {
            const int lane = (int)(threadIdx.x & 63u);
            for (uint32_t l = 0; l < 6u; ++l) {
                const unsigned long long m = __ballot(deferred && layer == l);
                if (!m) continue;
                const int leader = (int)__builtin_ctzll(m);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&ws.clip_count[l], (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, leader);
                if (deferred && layer == l) {
                    const uint32_t q = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (q < ws.slots) ws.clip_queue[(size_t)l * ws.slots + q] = t;
                }
            }
        }
    }
}

// The triangles k_setup queued: clipped against the near, far and guard-band planes, their fan
// triangles set up and counted into the bins.
template <bool SHADOW>
__global__ __launch_bounds__(64) void k_setup_clip(const float4* verts, const uint32_t* idx, RasterWS ws, int W, int H,
                                                   int tile_rank, int tile_world, ShadowXf xf) {
    latency_bound_kernel();
    const uint32_t layer = blockIdx.y;
    const uint32_t n = min(ws.clip_count[layer], ws.slots);
    SetupTarget tg = setup_target(ws, layer, W, H, tile_rank, tile_world);
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n; k += gridDim.x * 64u) {
        const uint32_t t = ws.clip_queue[(size_t)layer * ws.slots + k];
        V4 c[3];
        for (int j = 0; j < 3; ++j) c[j] = setup_vertex<SHADOW>(verts, idx[3 * t + j], layer, xf);
        emit_clipped<SHADOW>(c, t, t, ws, tg);
    }
}

// The two 12-triangle helper models.  mode 0: light marker (light_debug.wgsl:12-21), its
// records continue the geometry model's slots and ids; mode 1: fog volume (fog.wgsl:14-34)
// into its own record array.  enabled == 0 only invalidates the slots.
// Most of the fog volume's triangles need the clipper when the camera is inside or near the
// volume; its vertex arrays live in LDS here (in per-lane scratch memory the clipper's chains of
// dependent accesses made this 12-triangle kernel take as long as the set-up of the whole scene).
constexpr int kSmallTris = 16;  // triangles per workgroup (one wavefront)
__global__ __launch_bounds__(64) void k_setup_small(const float4* pos, const uint32_t* idx, uint32_t ntris, int mode, int enabled,
                                                    uint32_t slot_base, RasterWS ws, FrameConsts fc) {
    latency_bound_kernel();
    __shared__ ClipVert s_clip[kSmallTris][20];
    uint32_t t = blockIdx.x * kSmallTris + threadIdx.x;
    const bool mine = threadIdx.x < (uint32_t)kSmallTris && t < ntris;
    const bool have = mine && enabled;
    if (mine && !enabled) {
        ws.bb[slot_base + t] = make_uint2(kBBInvalid, kBBNothing);
    }
    V4 c[3];
    if (have)
        for (int k = 0; k < 3; ++k) {
            float4 p = pos[idx[3 * t + k]];
            V4 wp;
            if (mode == 0) wp = V4{p.x * 10.0f + fc.light_pos[0], p.y * 10.0f + fc.light_pos[1], p.z * 10.0f + fc.light_pos[2], 1.0f};
            else wp = mat4_mul(fc.model_fog, V4{p.x, p.y, p.z, 1.0f});
            c[k] = mat4_mul(fc.pv, wp);
        }
    SetupTarget tg = setup_target(ws, 0, fc.W, fc.H, fc.geom_tile_rank, fc.geom_tile_world);
    tg.count = mode == 0;
    emit_triangle<false, false>(have, c, slot_base + t, slot_base + t, ws, tg, s_clip[threadIdx.x & (kSmallTris - 1)]);
}

// ------------------------------------------------------------------ binning
// Bin counts are accumulated by the set-up kernels; after k_scan turned them into offsets
// this kernel appends every record index to the list of each bin it touches.  One thread
// per record slot, reading only the 8-byte bounding boxes; records inside one bin (almost
// all) are appended with wave-aggregated atomics.
__global__ __launch_bounds__(256) void k_bin_fill(RasterWS ws, int tile_rank, int tile_world) {
    latency_bound_kernel();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t layer = blockIdx.y;
    if (i == 0u && layer == 0u) *ws.total_items = *ws.scan_total;  // what the statistics report
    uint32_t ovf = ws.ovf_count[layer];
    if (ovf > ws.ovf_cap) ovf = ws.ovf_cap;
    const size_t stride = (size_t)ws.slots + ws.ovf_cap;
    uint2 bb = make_uint2(kBBInvalid, 0u);
    if (i < ws.slots + ovf) bb = ws.bb[layer * stride + i];
    const bool valid = bb.x != kBBInvalid;
    int b = 0;
    const bool single = valid && bb_single_bin(bb.x, bb.y, ws.bins_x, b);
    const uint32_t gb = layer * (uint32_t)(ws.bins_x * ws.bins_y) + (uint32_t)b;
    const bool go = single && bin_owned(b, tile_rank, tile_world);
    uint32_t pos = wave_agg_inc(ws.bin_cursor, go, gb);
    if (go) {
        uint32_t o = ws.bin_offset[gb] + pos;
        if (o < ws.items_cap) ws.bin_items[o] = i;
    }
    if (valid && !single) {
        Rec r = load_rec(&ws.recs[layer * stride + i]);
        bin_record_multi<true>(ws, layer, i, r, tile_rank, tile_world);
    }
}

// Gives every bin its range of the item list: bin_offset[b] = start of a range of bin_count[b]
// entries.  The ranges need not follow the order of the bins, so each workgroup scans its 256
// counts and takes the room for them with one atomic: two dozen independent workgroups instead of
// one that walks all the counts, on a path every frame waits for twice.
constexpr uint32_t kScanThreads = 256;
__global__ __launch_bounds__(kScanThreads) void k_scan(RasterWS ws) {
    latency_bound_kernel();
    __shared__ uint32_t part[kScanThreads / 64], s_base;
    const uint32_t n = (uint32_t)(ws.bins_x * ws.bins_y * ws.layers);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i = blockIdx.x * kScanThreads + tid;
    const uint32_t c = i < n ? ws.bin_count[i] : 0u;
    uint32_t inc = c;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, d);
        if ((int)lane >= d) inc += v;
    }
    if (lane == 63u) part[wave] = inc;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) before += part[w];
        total += part[w];
    }
    if (tid == 0u) {
        const uint32_t base = total ? atomicAdd(ws.scan_total, total) : 0u;
        s_base = base;
        if ((unsigned long long)base + total > ws.items_cap) atomicOr(ws.err, kErrItemOverflow);
    }
    __syncthreads();
    if (i < n) ws.bin_offset[i] = s_base + before + (inc - c);
}

// Exclusive prefix sum of the bin counts in one workgroup (the version for frames that overlap:
// it has to find room next to long-running kernels of other frames).
__global__ __launch_bounds__(kScanThreads) void k_scan_one(RasterWS ws) {
    latency_bound_kernel();
    __shared__ uint32_t part[kScanThreads];
    uint32_t n = (uint32_t)(ws.bins_x * ws.bins_y * ws.layers);
    uint32_t tid = threadIdx.x;
    uint32_t chunk = (n + kScanThreads - 1u) / kScanThreads;
    uint32_t beg = min(tid * chunk, n), end = min(beg + chunk, n);
    uint32_t s = 0;
    for (uint32_t k = beg; k < end; ++k) s += ws.bin_count[k];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t k = beg; k < end; ++k) {
        ws.bin_offset[k] = run;
        run += ws.bin_count[k];
    }
    if (tid == kScanThreads - 1u) {
        *ws.scan_total = part[kScanThreads - 1u];
        if (part[kScanThreads - 1u] > ws.items_cap) atomicOr(ws.err, kErrItemOverflow);
    }
}

// ------------------------------------------------------------------ rasteriser
// One workgroup per (bin, layer) resolves the bin in an LDS tile and writes it out once.
// kRasterShadow: depth-only, LessEqual with bias -> plain min of the biased depth
// (pass.rs:118-126).  kRasterColor: depth test Less with depth write (state.rs:467,482): the
// smallest (depth, id) key wins, ties go to the earlier draw.  kRasterLayers: second walk
// over the colour target's bins that hold pixels flagged for ordered blending; every
// fragment covering such a pixel is appended to the pixel's list.
//
// The bin's triangle list is consumed one workgroup's worth of records (THREADS, 256 in this
// build) at a time, one record per thread, all loads in flight together.  A triangle of at most 64x64 pixels with at most
// WR_RASTER_LANE_AREA pixels of its box inside the bin is rasterised by the thread that
// fetched it (64 triangles per wavefront in flight, 32-bit stepped edge functions); larger
// ones are queued in LDS and then taken one per wavefront, 64 lanes testing one 8x8 pixel
// block per step with 64-bit edge functions.
constexpr int kRasterColor = 0, kRasterShadow = 1, kRasterLayers = 2;
constexpr int kReplayK = 16;  // covering fragments kept per flagged pixel before the slow path

// Where k_raster<kRasterLayers> puts the fragments of pixels flagged for ordered blending.
struct LayerSink {
    const uint32_t* rp_slot;   // [H][W] list slot + 1 of a flagged pixel, else 0
    const uint32_t* vis;       // [H][W] id of the front-most fragment
    const uint32_t* bin_flag;  // [bins] non-zero: the bin holds a flagged pixel
    uint2* lists;              // [kReplayK][cap] (id, depth bits): entry c of slot s at c * cap + s
    uint32_t* list_count;      // [cap] fragments seen (may exceed kReplayK)
    uint32_t cap;
};

// A record every lane of the wavefront holds a copy of, moved to scalar registers: what is
// computed from it afterwards (edge set-up, block stepping, block rejection) then runs on
// the scalar unit.
__device__ inline Rec uniform_rec(const Rec& v) {
    Rec r;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Rec) / 4); ++i) dst[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)src[i]);
    return r;
}

__device__ inline int64_t uniform_i64(int64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Position of pixel (x, y) of a bin in its LDS tile.  The eight rows of an 8x8 pixel block
// are rotated to different column groups, so that a wavefront working on one block spreads
// over the LDS banks instead of hitting the same eight.
__device__ inline int tile_index(int x, int y) { return y * kBin + (x ^ ((y & 7) << 3)); }

// One covered sample.  Colour: depth test Less with depth write, the smallest (depth, id)
// key wins.  Shadow: min of the biased depth.  Layers: a flagged pixel's tile word holds
// its list slot + 1 (low half) and the id of its front-most fragment (high half), tile32
// counts the fragments appended so far.
template <int MODE>
__device__ inline void raster_emit(const Rec& r, float z, int li, unsigned long long* tile, uint32_t* tile32,
                                   const LayerSink& ls) {
    if (MODE == kRasterShadow) {
        float zb = fminf(fmaxf(z + r.iw0, 0.0f), 1.0f);
        atomicMin(&tile32[li], f2u(zb));
    } else if (MODE == kRasterColor) {
        if (z < 1.0f) atomicMin(&tile[li], ((unsigned long long)f2u(z) << 32) | (unsigned long long)r.id);
    } else {
        const unsigned long long w = tile[li];
        const uint32_t slot1 = (uint32_t)w;
        // a fragment drawn after the front-most one failed the depth test against it
        if (slot1 == 0u || r.id > (uint32_t)(w >> 32)) return;
        const uint32_t c = atomicAdd(&tile32[li], 1u);
        if (c < (uint32_t)kReplayK) ls.lists[(size_t)c * ls.cap + (slot1 - 1u)] = make_uint2(r.id, f2u(z));
    }
}

#ifndef WR_RASTER_LANE_AREA
#define WR_RASTER_LANE_AREA 128
#endif
#ifndef WR_RASTER_LANE_AREA_LAYERS
#define WR_RASTER_LANE_AREA_LAYERS 64
#endif
// Largest part of a bin (pixels of the clipped box) a triangle may cover and still be rasterised
// by the lane that fetched it.  The layers walk appends to lists in global memory for every
// covered flagged pixel, a lane's serial loop over its box is slower there, and the box at which
// a wavefront per triangle pays is smaller: 81 -> 63 us alone with 64 instead of 128 (76 with 32),
// while the colour and shadow walks are fastest with 128 (63 / 72 / 98 us and 162 / 165 / 168 us
// with 128 / 64 / 32).
template <int MODE>
constexpr int raster_lane_area() { return MODE == 2 ? WR_RASTER_LANE_AREA_LAYERS : WR_RASTER_LANE_AREA; }
// THREADS is chosen by the host per target; both targets use 256.  The shadow faces have 6144
// mostly short lists, where small workgroups keep more bins in flight per CU.  The colour
// target has few bins, some with thousands of triangles, where 1024 threads sharing the tile
// shorten the longest lists; measured, that gained less than the smaller groups' better
// overlap with the kernels of the neighbouring streams.
#ifdef WR_DEV
__device__ unsigned long long g_dbg_cycles[3 * 40000];
// Developer build: wall-clock ticks of thread 0 of every bin's workgroup, summed per raster mode and
// phase: [0] tile initialisation up to the first barrier, [1] list + record loads and the
// one-lane-per-triangle path, [2] the wavefront-per-triangle path, [3] write-out, [4] bins, [5] chunks,
// [6] large triangles.
__device__ unsigned long long g_dbg_phase[3][8];
// shadow walk, wavefront path: [0] 8x8 blocks visited, [1] of those, blocks in which no texel got nearer,
// [2] blocks in which every texel is covered; lane path: [3] triangles, [4] of those, triangles that changed no texel
__device__ unsigned long long g_dbg_blocks[8];
#define WR_PHASE(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_dbg_phase[MODE][i], t_ - dbg_tp); dbg_tp = t_; } } while (0)
#else
#define WR_PHASE(i) ((void)0)
#endif
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_raster(RasterWS ws, int W, int H, uint32_t* vis, float* depth,
                                                    int tile_rank, int tile_world, LayerSink ls) {
    latency_bound_kernel();
    constexpr int kRasterThreads = THREADS;
    __shared__ unsigned long long tile[MODE == kRasterShadow ? 1 : kBin * kBin];
    __shared__ uint32_t tile32[MODE == kRasterColor ? 1 : kBin * kBin];  // layers: fragments appended so far
    __shared__ uint32_t bigq[kRasterThreads];  // record indices of the chunk's large triangles
    __shared__ uint32_t s_nbig;
    __shared__ unsigned long long s_fmask;  // layers: bit (8*by + bx) set = 8x8 block holds a flagged pixel
    __shared__ uint32_t s_tmin[MODE == kRasterShadow ? 64 : 1], s_tmax[MODE == kRasterShadow ? 64 : 1];  // shadow: depth range per 8x8 tile
    const int b = blockIdx.x, layer = blockIdx.y;
#ifdef WR_DEV
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_tp = dbg_t0;
#endif
    if (!bin_owned(b, tile_rank, tile_world)) return;
    if (MODE == kRasterLayers && ls.bin_flag[b] == 0u) return;
    const int bx = b % ws.bins_x, by = b / ws.bins_x;
    const int ox = bx * kBin, oy = by * kBin;
    if (MODE == kRasterLayers) {
        if (threadIdx.x == 0) s_fmask = 0ull;
        __syncthreads();
    }
    if (MODE == kRasterShadow && threadIdx.x < 64) { s_tmin[threadIdx.x] = 0x7F7FFFFFu; s_tmax[threadIdx.x] = 0u; }
    for (int i = threadIdx.x; i < kBin * kBin; i += kRasterThreads) {
        if (MODE == kRasterShadow) tile32[i] = 0x3F800000u;
        else if (MODE == kRasterColor) tile[i] = 0x3F800000FFFFFFFFull;
        else {
            // a wavefront holds one 64-pixel row of the bin
            int px = ox + (i & (kBin - 1)), py = oy + (i / kBin);
            uint32_t slot = (px < W && py < H) ? ls.rp_slot[(size_t)py * W + px] : 0u;
            const uint32_t front = slot ? ls.vis[(size_t)py * W + px] : 0u;
            tile[tile_index(i & (kBin - 1), i / kBin)] = ((unsigned long long)front << 32) | slot;
            tile32[tile_index(i & (kBin - 1), i / kBin)] = 0u;
            unsigned long long m = __ballot(slot != 0u);
            m |= m >> 4; m |= m >> 2; m |= m >> 1;
            m &= 0x0101010101010101ull;
            const unsigned long long bits = (m * 0x0102040810204080ull) >> 56;  // one bit per 8 pixels
            if ((threadIdx.x & 63) == 0 && bits) atomicOr(&s_fmask, bits << (8 * ((i / kBin) >> 3)));
        }
    }
    const uint32_t gb = (uint32_t)layer * (uint32_t)(ws.bins_x * ws.bins_y) + (uint32_t)b;
    const uint32_t n = ws.bin_count[gb], off = ws.bin_offset[gb];
    const Rec* recs = ws.recs + (size_t)layer * ((size_t)ws.slots + ws.ovf_cap);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 7, ly = lane >> 3;
    for (uint32_t c0 = 0; c0 < n; c0 += kRasterThreads) {
        if (tid == 0) s_nbig = 0;
        __syncthreads();  // also orders the tile initialisation before the first atomics
        if (c0 == 0) WR_PHASE(0);
        if (c0 + tid < n && off + c0 + tid < ws.items_cap) {
            const uint32_t ri = ws.bin_items[off + c0 + tid];
            Rec r = load_rec(&recs[ri]);
            int x0 = imax((int)(r.bbmin & 0xffffu), ox), x1 = imin((int)(r.bbmax & 0xffffu), ox + kBin - 1);
            int y0 = imax((int)(r.bbmin >> 16), oy), y1 = imin((int)(r.bbmax >> 16), oy + kBin - 1);
            const int fw = imax(r.X0, imax(r.X1, r.X2)) - imin(r.X0, imin(r.X1, r.X2));
            const int fh = imax(r.Y0, imax(r.Y1, r.Y2)) - imin(r.Y0, imin(r.Y1, r.Y2));
            bool wanted = true;
            if (MODE == kRasterLayers) {
                // 8x8 blocks of the bin the triangle's box touches, against the flagged blocks
                wanted = false;
                if (x0 <= x1 && y0 <= y1) {
                    const int c0b = (x0 - ox) >> 3, c1b = (x1 - ox) >> 3, r0b = (y0 - oy) >> 3, r1b = (y1 - oy) >> 3;
                    const unsigned long long cols = ((1ull << (c1b - c0b + 1)) - 1ull) << c0b;
                    const int nrow = 8 * (r1b - r0b + 1);
                    const unsigned long long rows = (nrow >= 64 ? ~0ull : ((1ull << nrow) - 1ull)) << (8 * r0b);
                    wanted = ((cols * 0x0101010101010101ull) & rows & s_fmask) != 0ull;
                }
            }
            if (!wanted) {
            } else if (fw <= 64 * 256 && fh <= 64 * 256 && (x1 - x0 + 1) * (y1 - y0 + 1) <= raster_lane_area<MODE>()) {
                // The triangle's vertices span at most 64x64 pixels, so every edge function
                // stays below 2^30 over its box and can be stepped in 32-bit integers; the
                // values are the same integers the 64-bit evaluation gives.
                Edges e;
                make_edges(r, e);
                int64_t E[3];
                edge_eval(e, x0, y0, E);
                const int32_t dx0 = e.A[0] * 256, dx1 = e.A[1] * 256, dx2 = e.A[2] * 256;
                const int32_t dy0 = e.B[0] * 256, dy1 = e.B[1] * 256, dy2 = e.B[2] * 256;
                int32_t r0 = (int32_t)E[0] + e.tlb[0], r1 = (int32_t)E[1] + e.tlb[1], r2 = (int32_t)E[2] + e.tlb[2];
#ifdef WR_DEV
                bool dbg_any = false;
#endif
                for (int py = y0; py <= y1; ++py) {
                    int32_t b0 = r0, b1 = r1, b2 = r2;
                    for (int px = x0; px <= x1; ++px) {
                        if ((b0 | b1 | b2) >= 0) {
                            float l1 = (float)(b1 - e.tlb[1]) * r.inv2A, l2 = (float)(b2 - e.tlb[2]) * r.inv2A;
                            float z = (r.z0 + l1 * r.dz1) + l2 * r.dz2;
                            z = fminf(fmaxf(z, 0.0f), 1.0f);
#ifdef WR_DEV
                            if (MODE == kRasterShadow) dbg_any = dbg_any || f2u(fminf(fmaxf(z + r.iw0, 0.0f), 1.0f)) < tile32[tile_index(px - ox, py - oy)];
#endif
                            raster_emit<MODE>(r, z, tile_index(px - ox, py - oy), tile, tile32, ls);
                        }
                        b0 += dx0; b1 += dx1; b2 += dx2;
                    }
                    r0 += dy0; r1 += dy1; r2 += dy2;
                }
#ifdef WR_DEV
                if (MODE == kRasterShadow) { atomicAdd(&g_dbg_blocks[3], 1ull); if (!dbg_any) atomicAdd(&g_dbg_blocks[4], 1ull); }
#endif
            } else {
                bigq[atomicAdd(&s_nbig, 1u)] = ri;
            }
        }
        __syncthreads();
        WR_PHASE(1);
        const uint32_t nbig = s_nbig;
#ifdef WR_DEV
        if (threadIdx.x == 0) { atomicAdd(&g_dbg_phase[MODE][5], 1ull); atomicAdd(&g_dbg_phase[MODE][6], (unsigned long long)nbig); }
#endif
        for (uint32_t q = wave; q < nbig; q += kRasterThreads / 64) {
            // Wave-uniform record.  Step 1: lane (bx, by) classifies 8x8 pixel block (bx, by)
            // of the bin: M[i] is the largest value edge function i (plus its fill-rule
            // offset) takes on the block's 64 pixel centres; the block cannot touch the
            // triangle if any M[i] is negative.  Step 2: the blocks that can are visited one
            // by one, lane (lx, ly) testing pixel (lx, ly) of the block: its edge values are
            // the block's M[i] (fetched from the classifying lane) plus a per-lane offset
            // L[i] <= 0 fixed per triangle.  Rejected blocks cost nothing in step 2.
            const uint32_t ri = __builtin_amdgcn_readfirstlane(bigq[q]);
            const Rec r = uniform_rec(load_rec(&recs[ri]));
            Edges e;
            make_edges(r, e);
            const int x0 = imax((int)(r.bbmin & 0xffffu), ox), x1 = imin((int)(r.bbmax & 0xffffu), ox + kBin - 1);
            const int y0 = imax((int)(r.bbmin >> 16), oy), y1 = imin((int)(r.bbmax >> 16), oy + kBin - 1);
            int64_t L[3], Mv[3], Mo[3];
            edge_eval(e, ox, oy, Mo);
            for (int i = 0; i < 3; ++i) {
                const int64_t maxoff = (e.A[i] > 0 ? (int64_t)e.A[i] * 1792 : 0) + (e.B[i] > 0 ? (int64_t)e.B[i] * 1792 : 0);
                L[i] = ((int64_t)e.A[i] * (256 * lx) + (int64_t)e.B[i] * (256 * ly)) - maxoff;
                Mv[i] = (Mo[i] + maxoff + e.tlb[i]) + ((int64_t)e.A[i] * (2048 * lx) + (int64_t)e.B[i] * (2048 * ly));
            }
            const bool in_box = lx >= ((x0 - ox) >> 3) && lx <= ((x1 - ox) >> 3) && ly >= ((y0 - oy) >> 3) && ly <= ((y1 - oy) >> 3);
            unsigned long long todo = __ballot(in_box && (Mv[0] | Mv[1] | Mv[2]) >= 0);
            if (MODE == kRasterLayers) todo &= s_fmask;
            // blocks whose covered pixels have edge values below 2^30 (0 <= value <= M)
            const unsigned long long narrow = __ballot((Mv[1] | Mv[2]) < (1ll << 30));
            while (todo) {
                const int blk = (int)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                int64_t T[3];
#ifdef WR_DEV
                bool dbg_changed = false, dbg_cov = false;
#endif
                for (int i = 0; i < 3; ++i) {
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)Mv[i], blk);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)Mv[i] >> 32), blk);
                    T[i] = (int64_t)(((uint64_t)hi << 32) | lo) + L[i];
                }
                if ((T[0] | T[1] | T[2]) >= 0) {
                    float f1, f2;
                    if ((narrow >> blk) & 1ull) {
                        f1 = (float)((int32_t)T[1] - e.tlb[1]);
                        f2 = (float)((int32_t)T[2] - e.tlb[2]);
                    } else {
                        f1 = (float)(T[1] - e.tlb[1]);
                        f2 = (float)(T[2] - e.tlb[2]);
                    }
                    const float l1 = f1 * r.inv2A, l2 = f2 * r.inv2A;
                    float z = (r.z0 + l1 * r.dz1) + l2 * r.dz2;
                    z = fminf(fmaxf(z, 0.0f), 1.0f);
#ifdef WR_DEV
                    if (MODE == kRasterShadow) {
                        const float zb_ = fminf(fmaxf(z + r.iw0, 0.0f), 1.0f);
                        dbg_changed = f2u(zb_) < tile32[tile_index((blk & 7) * 8 + lx, (blk >> 3) * 8 + ly)];
                        dbg_cov = true;
                    }
#endif
                    raster_emit<MODE>(r, z, tile_index((blk & 7) * 8 + lx, (blk >> 3) * 8 + ly), tile, tile32, ls);
                }
#ifdef WR_DEV
                if (MODE == kRasterShadow) {
                    const unsigned long long ch = __ballot(dbg_changed), cv = __ballot(dbg_cov);
                    if (lane == 0) { atomicAdd(&g_dbg_blocks[0], 1ull); if (!ch) atomicAdd(&g_dbg_blocks[1], 1ull); if (cv == ~0ull) atomicAdd(&g_dbg_blocks[2], 1ull); }
                }
#endif
            }
        }
        __syncthreads();
        WR_PHASE(2);
    }
    __syncthreads();
    if (n == 0) WR_PHASE(0);
    if (MODE == kRasterShadow) {
        // The face's texels, and the depth range of each 8x8 texel tile behind the faces (see
        // ShadowMap): a wavefront holds one row of the bin, eight lanes one row of a tile.  Depths
        // are in [0, 1], where the order of the floats is the order of their bits.
        static_assert(kBin == 64 && kShadowTile == 8 && THREADS % 64 == 0, "one wavefront per bin row");
        for (int i = threadIdx.x; i < kBin * kBin; i += kRasterThreads) {
            const int x = i & (kBin - 1), y = i / kBin;
            const int px = ox + x, py = oy + y;
            const bool in = px < W && py < H;
            const uint32_t zb = tile32[tile_index(x, y)];
            if (in) depth[(size_t)layer * (size_t)W * (size_t)H + (size_t)py * W + px] = u2f(zb);
            uint32_t lo = in ? zb : 0x7F7FFFFFu, hi = in ? zb : 0u;
            for (int m = 1; m < 8; m <<= 1) {
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, m));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, m));
            }
            if ((x & 7) == 0 && lo != 0x7F7FFFFFu) {
                atomicMin(&s_tmin[(y >> 3) * 8 + (x >> 3)], lo);
                atomicMax(&s_tmax[(y >> 3) * 8 + (x >> 3)], hi);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int tx = (ox >> 3) + (int)(threadIdx.x & 7u), ty = (oy >> 3) + (int)(threadIdx.x >> 3);
            const int tw = shadow_tiles_w(W);
            if (tx < tw && ty < shadow_tiles_w(H)) {
                float* T = depth + (size_t)6 * (size_t)W * (size_t)H + 2 * ((size_t)layer * tw * tw + (size_t)ty * tw + tx);
                T[0] = u2f(s_tmin[threadIdx.x]);
                T[1] = u2f(s_tmax[threadIdx.x]);
            }
        }
#ifdef WR_DEV
        if (threadIdx.x == 0) g_dbg_cycles[MODE * 40000 + (int)gb] = wall_clock64() - dbg_t0;
        WR_PHASE(3);
        if (threadIdx.x == 0) atomicAdd(&g_dbg_phase[MODE][4], 1ull);
#endif
        return;
    }
    for (int i = threadIdx.x; i < kBin * kBin; i += kRasterThreads) {
        int px = ox + (i & (kBin - 1)), py = oy + (i / kBin);
        if (px >= W || py >= H) continue;
        size_t o = (size_t)layer * (size_t)W * (size_t)H + (size_t)py * W + px;
        const int ti = tile_index(i & (kBin - 1), i / kBin);
        if (MODE == kRasterLayers) {
            const uint32_t slot1 = (uint32_t)tile[ti];
            if (slot1 != 0u) ls.list_count[slot1 - 1u] = tile32[ti];
        } else {
            unsigned long long k = tile[ti];
            vis[o] = (uint32_t)k;
            depth[o] = u2f((uint32_t)(k >> 32));
        }
    }
#ifdef WR_DEV
    if (threadIdx.x == 0) g_dbg_cycles[MODE * 40000 + (int)gb] = wall_clock64() - dbg_t0;
    WR_PHASE(3);
    if (threadIdx.x == 0) atomicAdd(&g_dbg_phase[MODE][4], 1ull);
#endif
}

// ------------------------------------------------------------------ deferred shading
// Work space of the ordered-blend kernels (k_shade flags, k_raster<kRasterLayers> collects,
// k_resolve_shade orders, shades and blends).
struct BlendWS {
    uint32_t* rp_slot;     // [H][W] list slot + 1 of a flagged pixel, else 0
    uint32_t* slot_pixel;  // [cap] pixel index of a slot
    uint32_t* counts;      // [0] flagged pixels, [2] slow pixels; cleared every frame
    uint32_t* bin_flag;    // [bins] bin holds a flagged pixel; cleared every frame
    uint2* lists;          // [kReplayK][cap] (id, depth bits) of the fragments covering a flagged pixel
    uint32_t* list_count;  // [cap]
    uint32_t* slow_slots;  // [cap] slots left to k_replay_slow
    uint32_t cap;
};

struct ShadeArgs {
    FrameConsts fc;
    RasterWS ws;
    const uint32_t* vis;
    const float* depth;
    uint32_t* color;
    const uint32_t* idx;
    const uint32_t* tri_mat;
    uint32_t ntris_geom;
    const float4 *v0, *v1, *v2, *v3;
    const MatDesc* mats;
    const TexDesc* tex;
    const uint32_t* texels;
    const float* tables;
    const float* shadow;
    unsigned long long* counters;  // [0] fog fragments [1] noise evals the marches read [2] shaded fragments [3] replay pixels [4] slow-path pixels
                                   // [5] noise evals of k_fog_noise_view [6] of k_fog_noise_light
    BlendWS bw;
};

// Copies the colour tables (decode | unorm | thresh, 768 floats) to LDS; ends with a barrier.
__device__ inline void stage_color_tables(float* s_tab, const float* tables) {
    for (int i = (int)threadIdx.x; i < 768; i += (int)blockDim.x) s_tab[i] = tables[i];
    __syncthreads();
}

// Interpolates the VertexOutput of one fragment and runs pbr.wgsl's fs_main on it.
// Derivatives for the sampler's level of detail are taken per 2x2 quad on the fragment's
// own triangle: ddx = uv(x0+1,y0) - uv(x0,y0), ddy = uv(x0,y0+1) - uv(x0,y0).
// ALPHA_FIRST is for k_shade, which only keeps the colour of a fragment whose alpha is 1 (any
// other front-most fragment goes to the ordered-blend kernels, which shade it again in its
// place in the draw order): alpha is the base-colour texel's, so a fragment with another alpha
// returns after that one texture sample, colour unset.
template <bool ALPHA_FIRST>
__device__ inline V4 shade_fragment(const ShadeArgs& a, const float* color_tables, const Rec& r, const float* beta,
                                    const Edges& e, uint32_t tri, int px, int py, const int64_t* E) {
    uint32_t i0 = a.idx[3 * tri], i1 = a.idx[3 * tri + 1], i2 = a.idx[3 * tri + 2];
    float4 a0 = a.v0[i0], a1 = a.v0[i1], a2 = a.v0[i2];
    float B[3];
    bary_at(r, beta, E, B);
    PbrInputs in;
    in.uv = V2{ip(B, a0.x, a1.x, a2.x), ip(B, a0.y, a1.y, a2.y)};
    int qx = px & ~1, qy = py & ~1;
    V2 q[3];
    for (int k = 0; k < 3; ++k) {
        int64_t Eq[3];
        float Bq[3];
        edge_eval(e, qx + (k == 1), qy + (k == 2), Eq);
        bary_at(r, beta, Eq, Bq);
        q[k] = V2{ip(Bq, a0.x, a1.x, a2.x), ip(Bq, a0.y, a1.y, a2.y)};
    }
    in.ddx = V2{q[1].x - q[0].x, q[1].y - q[0].y};
    in.ddy = V2{q[2].x - q[0].x, q[2].y - q[0].y};
    ColorTables ct{color_tables};
    const MatDesc& m = a.mats[a.tri_mat[tri]];
    const V4 td = sample_material(a.texels, a.tex[m.base], ct, in.uv, in.ddx, in.ddy);
    if (ALPHA_FIRST && td.w != 1.0f) return V4{0.0f, 0.0f, 0.0f, td.w};
    float4 b0 = a.v1[i0], b1 = a.v1[i1], b2 = a.v1[i2];
    float4 c0 = a.v2[i0], c1 = a.v2[i1], c2 = a.v2[i2];
    float4 d0 = a.v3[i0], d1 = a.v3[i1], d2 = a.v3[i2];
    in.tpos = v3(ip(B, a0.z, a1.z, a2.z), ip(B, a0.w, a1.w, a2.w), ip(B, b0.x, b1.x, b2.x));
    in.tlight = v3(ip(B, b0.y, b1.y, b2.y), ip(B, b0.z, b1.z, b2.z), ip(B, b0.w, b1.w, b2.w));
    in.tview = v3(ip(B, c0.x, c1.x, c2.x), ip(B, c0.y, c1.y, c2.y), ip(B, c0.z, c1.z, c2.z));
    in.wpos = V4{ip(B, c0.w, c1.w, c2.w), ip(B, d0.x, d1.x, d2.x), ip(B, d0.y, d1.y, d2.y), ip(B, d0.z, d1.z, d2.z)};
    ShadowMap sm{a.shadow, a.fc.S};
    return fs_pbr(a.fc, sm, a.texels, a.tex, m, ct, in, td);
}

// Resolves (triangle id, fan index) to its record and clip weights.
__device__ inline Rec fetch_rec(const RasterWS& ws, uint32_t id, const float*& beta) {
    Rec r = load_rec(&ws.recs[id >> 3]);
    beta = nullptr;
    if (r.id == kRecClippedParent) {
        uint32_t ri = (uint32_t)r.X0 + (id & 7u);
        beta = ws.betas + (size_t)(ri - ws.slots) * 9;
        r = load_rec(&ws.recs[ri]);
    }
    return r;
}

// Ordered blending for a pixel whose front-most fragment is not opaque: replays, in draw
// order, every fragment that passed the depth test when it was drawn (geometry pass blends
// SrcAlpha/OneMinusSrcAlpha with depth writes on, state.rs:465-467; the light marker is
// drawn after it with REPLACE, state.rs:480-482).
__device__ __attribute__((noinline)) uint32_t replay_pixel(const ShadeArgs& a, int px, int py) {
    const RasterWS& ws = a.ws;
    ColorTables ct{a.tables};
    int b = (py / kBin) * ws.bins_x + (px / kBin);
    uint32_t n = ws.bin_count[b], off = ws.bin_offset[b];
    uint32_t col = 0xFF000000u;
    float d = 1.0f;
    long long last = -1;
    for (;;) {
        uint32_t best = 0xFFFFFFFFu, best_ri = 0;
        float best_z = 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t ri = ws.bin_items[off + i];
            const Rec* rp = &ws.recs[ri];
            uint32_t id = rp->id;
            if ((long long)id <= last || id >= best) continue;
            uint32_t bbmin = rp->bbmin, bbmax = rp->bbmax;
            if (px < (int)(bbmin & 0xffffu) || px > (int)(bbmax & 0xffffu) || py < (int)(bbmin >> 16) || py > (int)(bbmax >> 16)) continue;
            Rec r = load_rec(rp);
            Edges e;
            make_edges(r, e);
            int64_t E[3];
            edge_eval(e, px, py, E);
            if (!edges_inside(e, E)) continue;
            float z = depth_at(r, E);
            if (!(z < d)) continue;
            best = id; best_ri = ri; best_z = z;
        }
        if (best == 0xFFFFFFFFu) break;
        uint32_t tri = best >> 3;
        if (tri >= a.ntris_geom) {
            col = blend_pack(ct, col, V4{a.fc.light_color[0], a.fc.light_color[1], a.fc.light_color[2], 1.0f}, false);
        } else {
            Rec r = load_rec(&ws.recs[best_ri]);
            const float* beta = best_ri >= ws.slots ? ws.betas + (size_t)(best_ri - ws.slots) * 9 : nullptr;
            Edges e;
            make_edges(r, e);
            int64_t E[3];
            edge_eval(e, px, py, E);
            V4 src = shade_fragment<false>(a, a.tables, r, beta, e, tri, px, py, E);
            col = blend_pack(ct, col, src, true);
        }
        d = best_z;
        last = (long long)best;
    }
    return col;
}

template <bool COUNT>
// Six wavefronts per SIMD (at most 80 registers; the unconstrained allocation takes 88, i.e.
// five): shading waits on dependent fetches for half of a wavefront's life, and the sixth
// resident wavefront shortens the launch by 4 %; seven or eight would spill.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_shade(ShadeArgs a) {
    latency_bound_kernel();
    __shared__ uint32_t s_cnt, s_base;
    __shared__ float s_tab[768];
    // 16x16 pixels per workgroup, each wavefront an 8x8 block
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int bin = (py / kBin) * a.ws.bins_x + (px / kBin);  // the same for the whole workgroup
    if (a.fc.tile_world > 1 && (bin % a.fc.tile_world) != a.fc.tile_rank) return;
    const bool live = px < a.fc.W && py < a.fc.H;
    const size_t pi = (size_t)py * a.fc.W + px;
    const uint32_t id = live ? a.vis[pi] : kVisNone;
    stage_color_tables(s_tab, a.tables);
    ColorTables ct{s_tab};
    bool flagged = false;
    if (id == kVisNone) {
        if (live) a.color[pi] = 0xFF000000u;  // clear colour (0,0,0,1), state.rs:696-701
    } else if ((id >> 3) >= a.ntris_geom) {  // light marker, light_debug.wgsl:23-26
        a.color[pi] = blend_pack(ct, 0u, V4{a.fc.light_color[0], a.fc.light_color[1], a.fc.light_color[2], 1.0f}, false);
    } else {
        const uint32_t tri = id >> 3;
        const float* beta;
        Rec r = fetch_rec(a.ws, id, beta);
        Edges e;
        make_edges(r, e);
        int64_t E[3];
        edge_eval(e, px, py, E);
        V4 src = shade_fragment<true>(a, s_tab, r, beta, e, tri, px, py, E);
        if (COUNT) atomicAdd(&a.counters[2], 1ull);
        if (src.w == 1.0f) {
            a.color[pi] = blend_pack(ct, 0u, src, false);  // alpha 1 replaces whatever was blended before
        } else {
            // the front-most fragment is not opaque: this pixel's colour depends on draw order;
            // flag it for the ordered-blend kernels
            if (COUNT) atomicAdd(&a.counters[3], 1ull);
            flagged = true;
        }
    }
    if (!__syncthreads_or(flagged ? 1 : 0)) {
        if (live) a.bw.rp_slot[pi] = 0u;
        return;
    }
    const uint32_t slot = group_append(&a.bw.counts[0], flagged ? 1u : 0u, &s_cnt, &s_base);
    if (live) a.bw.rp_slot[pi] = flagged ? slot + 1u : 0u;
    if (flagged) a.bw.slot_pixel[slot] = (uint32_t)pi;
    if (threadIdx.x == 0) a.bw.bin_flag[bin] = 1u;
}

// Ordered blending, steps 2 to 4 (step 1 is k_raster<kRasterLayers>): one thread per flagged
// pixel.  Sorted by triangle id the collected fragments give the draw order, and a running
// minimum over depth gives those that passed the depth test when drawn.  The ones from the
// last fragment whose material cannot be translucent onwards (it replaced everything under
// it) are shaded one after the other (pbr.wgsl's fs_main; the light marker, drawn with
// REPLACE, has a constant colour: light_debug.wgsl:23-26) and blended over the clear colour in
// draw order (geometry: SrcAlpha/OneMinusSrcAlpha, state.rs:465-467; light marker: REPLACE,
// state.rs:480-482).  Pixels covered by more than kReplayK fragments are queued for
// k_replay_slow.  (Three kernels once: ordering, one lane per fragment to shade, blending.  In
// one, a frame's chain is two launches and 40 us shorter and 48 bytes per pixel of buffers are
// gone; that a lane now shades its pixel's two or three fragments in turn costs less.)
__global__ __launch_bounds__(256) void k_resolve_shade(ShadeArgs a) {
    latency_bound_kernel();
    __shared__ float s_tab[768];
    __shared__ uint32_t s_ids[kReplayK][256];  // the fragments a thread will shade, in draw order
    const int tid = threadIdx.x;
    const BlendWS& bw = a.bw;
    const uint32_t nslots = min(bw.counts[0], bw.cap);
    if (blockIdx.x * 256u >= nslots) return;
    stage_color_tables(s_tab, a.tables);
    ColorTables ct{s_tab};
    for (uint32_t s = blockIdx.x * 256u + (uint32_t)tid; s < nslots; s += gridDim.x * 256u) {
        const uint32_t cnt = bw.list_count[s];
        if (cnt > (uint32_t)kReplayK) {
            bw.slow_slots[atomicAdd(&bw.counts[2], 1u)] = s;
            continue;
        }
        uint32_t nkeep = 0u;
        {
            uint32_t id[kReplayK];
            float z[kReplayK];
#pragma unroll
            for (int i = 0; i < kReplayK; ++i) {
                const uint2 q = (uint32_t)i < cnt ? bw.lists[(size_t)i * bw.cap + s] : make_uint2(0xFFFFFFFFu, 0u);
                id[i] = q.x; z[i] = u2f(q.y);
            }
            static_assert(kReplayK == 16, "the sorting network below is the one for 16 keys");
#define CX(x, y)                                                                  \
    {                                                                             \
        const bool sw = id[x] > id[y];                                            \
        const uint32_t ia = sw ? id[y] : id[x], ib = sw ? id[x] : id[y];          \
        const float za = sw ? z[y] : z[x], zb = sw ? z[x] : z[y];                 \
        id[x] = ia; id[y] = ib; z[x] = za; z[y] = zb;                             \
    }
            CX(0, 1) CX(2, 3) CX(4, 5) CX(6, 7) CX(8, 9) CX(10, 11) CX(12, 13) CX(14, 15)
            CX(0, 2) CX(1, 3) CX(4, 6) CX(5, 7) CX(8, 10) CX(9, 11) CX(12, 14) CX(13, 15)
            CX(1, 2) CX(5, 6) CX(9, 10) CX(13, 14) CX(0, 4) CX(1, 5) CX(2, 6) CX(3, 7)
            CX(8, 12) CX(9, 13) CX(10, 14) CX(11, 15) CX(2, 4) CX(3, 5) CX(10, 12) CX(11, 13)
            CX(1, 2) CX(3, 4) CX(5, 6) CX(9, 10) CX(11, 12) CX(13, 14) CX(0, 8) CX(1, 9)
            CX(2, 10) CX(3, 11) CX(4, 12) CX(5, 13) CX(6, 14) CX(7, 15) CX(4, 8) CX(5, 9)
            CX(6, 10) CX(7, 11) CX(2, 4) CX(3, 5) CX(6, 8) CX(7, 9) CX(10, 12) CX(11, 13)
            CX(1, 2) CX(3, 4) CX(5, 6) CX(7, 8) CX(9, 10) CX(11, 12) CX(13, 14)
#undef CX
            uint32_t keep = 0u;
            float d = 1.0f;
#pragma unroll
            for (int i = 0; i < kReplayK; ++i)
                if ((uint32_t)i < cnt && z[i] < d) { d = z[i]; keep |= 1u << i; }
            int first_kept = kReplayK;
            bool found = false;
#pragma unroll
            for (int i = kReplayK - 1; i >= 0; --i)
                if (!found && ((keep >> i) & 1u)) {
                    const uint32_t tri = id[i] >> 3;
                    first_kept = i;
                    if (tri >= a.ntris_geom || a.mats[a.tri_mat[tri]].opaque) found = true;
                }
#pragma unroll
            for (int i = 0; i < kReplayK; ++i)
                if (i >= first_kept && ((keep >> i) & 1u)) s_ids[nkeep++][tid] = id[i];
        }
        const uint32_t pi = bw.slot_pixel[s];
        const int px = (int)(pi % (uint32_t)a.fc.W), py = (int)(pi / (uint32_t)a.fc.W);
        uint32_t col = 0xFF000000u;
        for (uint32_t k = 0; k < nkeep; ++k) {
            const uint32_t fid = s_ids[k][tid];
            const uint32_t tri = fid >> 3;
            V4 src = V4{a.fc.light_color[0], a.fc.light_color[1], a.fc.light_color[2], 1.0f};
            if (tri < a.ntris_geom) {
                const float* beta;
                Rec r = fetch_rec(a.ws, fid, beta);
                Edges e;
                make_edges(r, e);
                int64_t E[3];
                edge_eval(e, px, py, E);
                src = shade_fragment<false>(a, s_tab, r, beta, e, tri, px, py, E);
            }
            col = blend_pack(ct, col, src, tri < a.ntris_geom);
        }
        a.color[pi] = col;
    }
}

// Ordered blending of the pixels k_resolve_shade queued: the general, slow method.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_replay_slow(ShadeArgs a) {
    latency_bound_kernel();
    const BlendWS& bw = a.bw;
    const uint32_t n = min(bw.counts[2], bw.cap);
    if (blockIdx.x * 64u >= n) return;
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n; k += gridDim.x * 64u) {
        const uint32_t pi = bw.slot_pixel[bw.slow_slots[k]];
        if (COUNT) atomicAdd(&a.counters[4], 1ull);
        a.color[pi] = replay_pixel(a, (int)(pi % (uint32_t)a.fc.W), (int)(pi / (uint32_t)a.fc.W));
    }
}

// ------------------------------------------------------------------ fog
struct FogArgs {
    FrameConsts fc;
    RasterWS ws;          // fog volume records (no bins)
    const float4* pos;    // fog model positions
    const uint32_t* idx;
    uint32_t ntris;
    const float* depth;   // geometry depth: depth test (Less, no write) and fog.wgsl:131
    uint32_t* color;
    const float* shadow;
    const float* tables;
    unsigned long long* counters;
    uint32_t* item_count;  // march work items appended this frame
    uint32_t* item_pixel;  // [W*H] pixel index of each item
    float4* item_ray;      // [W*H] ray origin xyz + max_fog_depth
    NoiseTables tables_noise;  // hash tables of the low octaves (k_fog_tables)
    float* noise_view;     // [kFogMaxSteps][cap] fog_noise of view-march step s of item k at s * cap + k
    float* noise_light;    // [kFogLightMaxSteps][cap] the same for the light march, indexed by position in the lit list
    float4* item_res;      // [cap] view march of an item: density, depth, in_light, position in the lit list (bits);
                           // with k_fog_march: the item's fog colour
    uint32_t* lit_count;   // items whose fog point is lit (they need the light march)
    uint32_t* lit_item;    // [cap] their item indices
    uint32_t cap;          // W*H
    uint32_t* multi_count; // pixels with several fog fragments, finished by k_fog_multi
    uint32_t* multi_pixel; // [W*H]
};

// One fog fragment: interpolate the world position, run the pre-march part of fog.wgsl's
// fs_main.  Returns false if the fragment fails the depth test.
__device__ inline bool fog_fragment(const FogArgs& a, const Rec& r, const float* beta, uint32_t tri, int px, int py,
                                    float zbuf, V3& origin, float& max_depth) {
    Edges e;
    make_edges(r, e);
    int64_t E[3];
    edge_eval(e, px, py, E);
    if (!edges_inside(e, E)) return false;
    float z = depth_at(r, E);
    if (!(z < zbuf)) return false;  // depth test Less, no depth write (state.rs:497-502)
    float B[3];
    bary_at(r, beta, E, B);
    V4 w[3];
    for (int k = 0; k < 3; ++k) {
        float4 p = a.pos[a.idx[3 * tri + k]];
        w[k] = mat4_mul(a.fc.model_fog, V4{p.x, p.y, p.z, 1.0f});
    }
    origin = v3(ip(B, w[0].x, w[1].x, w[2].x), ip(B, w[0].y, w[1].y, w[2].y), ip(B, w[0].z, w[1].z, w[2].z));
    max_depth = fog_max_depth(a.fc, a.depth, z, px, py);
    return true;
}

// Edge functions and depth plane of one record, staged in LDS for a 16x16 pixel block.
struct StagedRec {
    int32_t A[3], B[3];
    int32_t tlb;      // bit i set: edge i is not top-left
    uint32_t ri;      // record index
    int64_t C[3];
    float z0, dz1, dz2, inv2A;
    uint32_t bbmin, bbmax;
    uint32_t id, pad;
};
static_assert(sizeof(StagedRec) == 88, "StagedRec layout");

// Fog pass, part 1 (state.rs:763-804): rasterises the fog volume per pixel.  A workgroup of 256
// threads owns a 32x32 pixel block (four wavefronts find room on a busy device long before the
// sixteen of a thread per pixel do, and one list append per block is all the single-address
// atomics the launch can afford): its threads first fetch the volume's records, one each, and
// stage those that can touch the block in LDS (edge set-up done once per record, not once
// per pixel); then every pixel tests the staged records.  A pixel with exactly one fog
// fragment (always the case for a convex volume) whose march is not empty becomes a work
// item for the march kernels, appended with one atomic per workgroup; a pixel with several
// fragments is queued for k_fog_multi.  Reads the geometry depth only, not the colour, so the
// fog pass up to and including the march can run next to the shading kernels.
constexpr int kFogBlock = 32;
constexpr int kFogSetupThreads = 256;  // a thread takes a 2x2 pixel quad
template <bool COUNT>
__global__ __launch_bounds__(kFogSetupThreads) void k_fog_setup(FogArgs a) {
    latency_bound_kernel();
    // Records are staged 64 at a time (5.6 KB): a fog volume has a dozen triangles, and a workgroup
    // that asks for little LDS finds a place on a compute unit whose LDS the shadow rasteriser's
    // tiles have all but filled.
    constexpr uint32_t kStage = 64;
    __shared__ StagedRec srec[kStage];
    __shared__ uint32_t s_m, s_cnt, s_base;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bx0 = blockIdx.x * kFogBlock, by0 = blockIdx.y * kFogBlock;
    // A wavefront owns a 16x16 quadrant of the block; its lanes go through the quadrant's four 8x8
    // tiles one after the other, sixteen lanes (4x4 quads) per tile, so that items appended in
    // lane order keep the pixels of a tile together: 64 consecutive items are one 8x8 tile.
    const int tile = lane >> 4, quad = lane & 15;
    const int qx = bx0 + (wave & 1) * 16 + (tile & 1) * 8 + (quad & 3) * 2;
    const int qy = by0 + (wave >> 1) * 16 + (tile >> 1) * 8 + (quad >> 2) * 2;
    if (a.fc.tile_world > 1) {  // a block lies in one bin
        int b = (by0 / kBin) * ((a.fc.W + kBin - 1) / kBin) + (bx0 / kBin);
        if ((b % a.fc.tile_world) != a.fc.tile_rank) return;
    }
    const int bx1 = imin(bx0 + kFogBlock - 1, a.fc.W - 1), by1 = imin(by0 + kFogBlock - 1, a.fc.H - 1);
    bool live[4];
    float zbuf[4];
    int nfrag[4];
    uint32_t hit_ri[4];
    int64_t hit_E[4][3];
    float hit_z[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int px = qx + (p & 1), py = qy + (p >> 1);
        live[p] = px < a.fc.W && py < a.fc.H;
        zbuf[p] = live[p] ? a.depth[(size_t)py * a.fc.W + px] : 0.0f;
        nfrag[p] = 0; hit_ri[p] = 0u; hit_z[p] = 0.0f;
        hit_E[p][0] = hit_E[p][1] = hit_E[p][2] = 0;
    }
    const uint32_t nrec = a.ws.slots + min(a.ws.ovf_count[0], a.ws.ovf_cap);
    for (uint32_t c0 = 0; c0 < nrec; c0 += kStage) {
        if (tid == 0) s_m = 0;
        __syncthreads();
        if ((uint32_t)tid < kStage && c0 + tid < nrec) {
            const Rec* rp = &a.ws.recs[c0 + tid];
            const uint2 bb = a.ws.bb[c0 + tid];
            const uint32_t bbmin = bb.x, bbmax = bb.y;
            if (bbmin != kBBInvalid &&
                !((int)(bbmin & 0xffffu) > bx1 || (int)(bbmax & 0xffffu) < bx0 || (int)(bbmin >> 16) > by1 || (int)(bbmax >> 16) < by0)) {
                Rec r = load_rec(rp);
                Edges e;
                make_edges(r, e);
                if (!box_rejected(e, bx0, by0, bx1, by1)) {
                    StagedRec& q = srec[atomicAdd(&s_m, 1u)];
                    for (int i = 0; i < 3; ++i) { q.A[i] = e.A[i]; q.B[i] = e.B[i]; q.C[i] = e.C[i]; }
                    q.tlb = (e.tlb[0] & 1) | (e.tlb[1] & 2) | (e.tlb[2] & 4);
                    q.ri = c0 + tid; q.id = r.id; q.z0 = r.z0; q.dz1 = r.dz1; q.dz2 = r.dz2; q.inv2A = r.inv2A;
                    q.bbmin = r.bbmin; q.bbmax = r.bbmax;
                }
            }
        }
        __syncthreads();
        const int m = (int)s_m;
        for (int j = 0; j < m; ++j) {
            const StagedRec& q = srec[j];
            // the quad against the record's box
            if (qx + 1 < (int)(q.bbmin & 0xffffu) || qx > (int)(q.bbmax & 0xffffu) || qy + 1 < (int)(q.bbmin >> 16) || qy > (int)(q.bbmax >> 16)) continue;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int px = qx + (p & 1), py = qy + (p >> 1);
                if (!live[p] || px < (int)(q.bbmin & 0xffffu) || px > (int)(q.bbmax & 0xffffu) || py < (int)(q.bbmin >> 16) || py > (int)(q.bbmax >> 16)) continue;
                const int32_t sx = px * 256 + 128, sy = py * 256 + 128;
                int64_t E0 = (int64_t)q.A[0] * sx + (int64_t)q.B[0] * sy + q.C[0];
                int64_t E1 = (int64_t)q.A[1] * sx + (int64_t)q.B[1] * sy + q.C[1];
                int64_t E2 = (int64_t)q.A[2] * sx + (int64_t)q.B[2] * sy + q.C[2];
                if (((E0 - (q.tlb & 1)) | (E1 - ((q.tlb >> 1) & 1)) | (E2 - ((q.tlb >> 2) & 1))) < 0) continue;
                float l1 = (float)E1 * q.inv2A, l2 = (float)E2 * q.inv2A;
                float z = (q.z0 + l1 * q.dz1) + l2 * q.dz2;
                z = fminf(fmaxf(z, 0.0f), 1.0f);
                if (!(z < zbuf[p])) continue;  // depth test Less, no depth write (state.rs:497-502)
                hit_ri[p] = q.ri; hit_E[p][0] = E0; hit_E[p][1] = E1; hit_E[p][2] = E2; hit_z[p] = z;
                if ((q.id >> 3) < a.ntris) ++nfrag[p];  // (always true for records written this frame)
            }
        }
        __syncthreads();
    }
    V3 origin[4];
    float max_depth[4];
    uint32_t want = 0u, multi = 0u, single = 0u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        origin[p] = v3(0.0f, 0.0f, 0.0f);
        max_depth[p] = 0.0f;
        if (nfrag[p] == 1) {
            // the pre-march part of fog.wgsl's fs_main on the interpolated world position
            Rec r = load_rec(&a.ws.recs[hit_ri[p]]);
            const float* beta = hit_ri[p] >= a.ws.slots ? a.ws.betas + (size_t)(hit_ri[p] - a.ws.slots) * 9 : nullptr;
            float B[3];
            bary_at(r, beta, hit_E[p], B);
            const uint32_t tri = r.id >> 3;
            V4 w[3];
            for (int k = 0; k < 3; ++k) {
                float4 v = a.pos[a.idx[3 * tri + k]];
                w[k] = mat4_mul(a.fc.model_fog, V4{v.x, v.y, v.z, 1.0f});
            }
            origin[p] = v3(ip(B, w[0].x, w[1].x, w[2].x), ip(B, w[0].y, w[1].y, w[2].y), ip(B, w[0].z, w[1].z, w[2].z));
            max_depth[p] = fog_max_depth(a.fc, a.depth, hit_z[p], qx + (p & 1), qy + (p >> 1));
            ++single;
        }
        if (nfrag[p] == 1 && max_depth[p] > 0.0f) ++want;
        if (nfrag[p] >= 2) ++multi;
    }
    if (COUNT && single) atomicAdd(&a.counters[0], (unsigned long long)single);
    if (__syncthreads_or(multi ? 1 : 0)) {
        // several fog fragments on one pixel (a fog volume that is not convex): left to k_fog_multi
        uint32_t m = group_append(a.multi_count, multi, &s_cnt, &s_base);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (nfrag[p] >= 2) a.multi_pixel[m++] = (uint32_t)((size_t)(qy + (p >> 1)) * a.fc.W + (qx + (p & 1)));
    }
    if (!__syncthreads_or(want ? 1 : 0)) return;
    uint32_t k = group_append(a.item_count, want, &s_cnt, &s_base);
#pragma unroll
    for (int p = 0; p < 4; ++p)
        if (nfrag[p] == 1 && max_depth[p] > 0.0f) {
            a.item_pixel[k] = (uint32_t)((size_t)(qy + (p >> 1)) * a.fc.W + (qx + (p & 1)));
            a.item_ray[k] = make_float4(origin[p].x, origin[p].y, origin[p].z, max_depth[p]);
            ++k;
        }
}

// Fog pass for the pixels k_fog_setup found covered by several fog fragments: every fragment
// that passes the depth test is marched and blended over the pixel in draw order.  Runs after
// the geometry colour is final.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_fog_multi(FogArgs a) {
    latency_bound_kernel();
    const uint32_t n = *a.multi_count;
    if (blockIdx.x * 64u >= n) return;
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n; k += gridDim.x * 64u) {
        const uint32_t pi = a.multi_pixel[k];
        const int px = (int)(pi % (uint32_t)a.fc.W), py = (int)(pi / (uint32_t)a.fc.W);
        const float zbuf = a.depth[pi];
        ColorTables ct{a.tables};
        ShadowMap sm{a.shadow, a.fc.S};
        uint32_t col = a.color[pi];
        for (uint32_t t = 0; t < a.ntris; ++t) {
            const uint2 sbb = a.ws.bb[t];
            if (sbb.x == kBBInvalid && sbb.y != kBBClippedParent) continue;
            uint32_t first = t, count = 1;
            if (sbb.x == kBBInvalid) {
                Rec parent = load_rec(&a.ws.recs[t]);
                first = (uint32_t)parent.X0; count = (uint32_t)parent.Y0;
            }
            for (uint32_t s = 0; s < count; ++s) {
                uint32_t ri = first + s;
                if (a.ws.bb[ri].x == kBBInvalid) continue;
                Rec r = load_rec(&a.ws.recs[ri]);
                if (px < (int)(r.bbmin & 0xffffu) || px > (int)(r.bbmax & 0xffffu) || py < (int)(r.bbmin >> 16) || py > (int)(r.bbmax >> 16)) continue;
                const float* beta = ri >= a.ws.slots ? a.ws.betas + (size_t)(ri - a.ws.slots) * 9 : nullptr;
                V3 o;
                float md;
                if (!fog_fragment(a, r, beta, t, px, py, zbuf, o, md)) continue;
                uint32_t evals = 0;
                V4 src = md <= 0.0f ? V4{0.0f, 0.0f, 0.0f, 0.0f} : fog_march_shade(a.fc, sm, nullptr, o, md, evals);
                if (COUNT) { atomicAdd(&a.counters[0], 1ull); atomicAdd(&a.counters[1], (unsigned long long)evals); }
                col = blend_pack(ct, col, src, true);
            }
        }
        a.color[pi] = col;
    }
}

// Fills the hash tables: entry (i,j,k) of a level = hash of lattice point lo + (i,j,k).  The
// workgroups of all levels form one grid.  A workgroup covers a tile of kTableSlices z slices:
// 2^xshift consecutive x (the power of two that holds the row, at most 256) by 256 >> xshift
// consecutive y, so narrow tables still fill their wavefronts and a thread decodes its index
// with shifts only; it then walks its column through the slices.  (One entry per thread made
// the launch a matter of workgroup dispatch rate: 27000 workgroups of 0.1 us of work each.)
constexpr int kTableSlices = 8;  // z slices one workgroup fills: a thread fills a column of that many entries
struct TablePlan {
    uint32_t fast_sine;     // bit l: every hash argument of level l is within the fast sine's range
    uint32_t wg_begin[17];  // first workgroup of each level (level l owns [wg_begin[l], wg_begin[l+1]))
    uint32_t xchunks[16];   // tiles per row = ceil(nx / 2^xshift)
    uint32_t ychunks[16];   // tiles per column = ceil(ny / (256 >> xshift))
    uint32_t xshift[16];
};
__host__ inline void plan_table_level(TablePlan& plan, int l, const NoiseTable& t) {
    uint32_t sh = 0;
    while (sh < 8 && (1u << sh) < (uint32_t)t.nx) ++sh;
    const uint32_t xw = 1u << sh, yr = 256u >> sh;
    plan.xshift[l] = sh;
    plan.xchunks[l] = ((uint32_t)t.nx + xw - 1u) / xw;
    plan.ychunks[l] = ((uint32_t)t.ny + yr - 1u) / yr;
    const uint32_t zchunks = ((uint32_t)t.nz + (uint32_t)kTableSlices - 1u) / (uint32_t)kTableSlices;
    plan.wg_begin[l + 1] = plan.wg_begin[l] + plan.xchunks[l] * plan.ychunks[l] * zchunks;
    // the largest hash argument of the level, with room for its three roundings
    const double ax = std::max(std::abs((double)t.lox), std::abs((double)t.lox + t.nx));
    const double ay = std::max(std::abs((double)t.loy), std::abs((double)t.loy + t.ny));
    const double az = std::max(std::abs((double)t.loz), std::abs((double)t.loz + t.nz));
    if ((ax * 12.345 + ay * 67.89 + az * 412.12) * 1.000001 < (double)kSinFastRange) plan.fast_sine |= 1u << l;
}
template <int TIER>
__device__ inline void fill_table_column(float* out, size_t slice, float xy, int z0, int count) {
    for (int k = 0; k < count; ++k) out[(size_t)k * slice] = hash_from_dot<TIER>(xy + (float)(z0 + k) * 412.12f);
}
__global__ __launch_bounds__(256) void k_fog_tables(NoiseTables nt, TablePlan plan) {
    const uint32_t b = blockIdx.x;
    int level = 0;
    while (level < 15 && b >= plan.wg_begin[level + 1]) ++level;
    const NoiseTable t = nt.level[level];
    if (t.data == nullptr || b >= plan.wg_begin[level + 1]) return;
    const uint32_t local = b - plan.wg_begin[level];
    const uint32_t xc = local % plan.xchunks[level], rest = local / plan.xchunks[level];
    const uint32_t yc = rest % plan.ychunks[level];
    const int iz0 = (int)(rest / plan.ychunks[level]) * kTableSlices;
    const uint32_t sh = plan.xshift[level];
    const int ix = (int)((xc << sh) + (threadIdx.x & ((1u << sh) - 1u)));
    const int iy = (int)(yc * (256u >> sh) + (threadIdx.x >> sh));
    if (ix >= t.nx || iy >= t.ny) return;
    const size_t slice = (size_t)t.nx * t.ny;
    float* out = const_cast<float*>(t.data) + ((size_t)iz0 * t.ny + iy) * t.nx + ix;
    // rand(u) of noise.wgsl:3-5 for the integer u = lo + (ix, iy, iz): (x*12.345 + y*67.89) + z*412.12
    const float xy = (float)(t.lox + ix) * 12.345f + (float)(t.loy + iy) * 67.89f;
    const int count = imin(kTableSlices, t.nz - iz0);
    if ((plan.fast_sine >> level) & 1u) fill_table_column<kSinFast>(out, slice, xy, t.loz + iz0, count);
    else fill_table_column<kSinGuarded>(out, slice, xy, t.loz + iz0, count);
}

// Fog pass, part 2: the marches of fog.wgsl:64-116, cut at their only sequential dependence.
// The steps of a march depend on each other through two scalars only (the running density and
// the loop exits); the expensive part, fog_noise at the step's position, depends on the step
// index alone, because `depth` is step_size added i + 1 times (march_step_depth).  So the noise
// of every step the depth exit allows is evaluated first, one (item, step) pair per lane, in
// uniform work units that fill the chip evenly; the running sums are then formed from the
// stored values by the loop of the shader itself (ray_march_stored).  A step that the density
// exit (density >= 1) would have skipped is evaluated and not read: that needs more than
// fifteen steps of noise above one half and is rare.
//   k_fog_noise_view   fog_noise of the view-march steps          (needs geometry depth only)
//   k_fog_view_resolve running sums, fog point, its shadow lookup (needs the shadow maps);
//                      lit fog points are listed for the light march
//   k_fog_noise_light  fog_noise of the light-march steps of lit fog points
//   k_fog_shade_blend  occlusion sums, fog colour, blend over the finished geometry colour
#ifndef WR_FOG_STEP_GROUP
#define WR_FOG_STEP_GROUP 3
#endif
constexpr int kFogStepGroup = WR_FOG_STEP_GROUP;  // march steps one wavefront evaluates for its 64 items
constexpr int kFogViewGroups = (kFogMaxSteps + kFogStepGroup - 1) / kFogStepGroup;
constexpr int kFogLightGroups = (kFogLightMaxSteps + kFogStepGroup - 1) / kFogStepGroup;

__device__ inline V3 fog_item_dir(const FrameConsts& fc, V3 origin) {  // fog.wgsl:127
    return normalize3(origin - v3(fc.cam_pos[0], fc.cam_pos[1], fc.cam_pos[2]));
}

// One wavefront: the noise of up to kFogStepGroup march steps of 64 consecutive items.
// Positions are origin + dir * march_step_depth(step_size, step); `out` points at the lane's
// column of the [step][cap] noise array.
template <bool COUNT>
__device__ inline void fog_noise_steps(const FogArgs& a, bool live, V3 origin, V3 dir, float max_depth, float step_size,
                                       int max_steps, int s0, float* out, unsigned long long* counter) {
#pragma unroll 1
    for (int s = s0; s < s0 + kFogStepGroup && s < max_steps; ++s) {
        const float depth = march_step_depth(step_size, s);
        const bool act = live && !(depth >= max_depth);  // the march's exit test, fog.wgsl:70-72
        if (!__any(act ? 1 : 0)) break;                  // depth only grows: no later step is active either
        if (act) out[(size_t)s * a.cap] = fog_noise_tab(a.fc, a.tables_noise, origin + dir * depth);
        if (COUNT && act) atomicAdd(counter, 1ull);
    }
}

// The number of items is only known on the device, so the grid is sized for the worst case
// (every pixel an item).  Workgroup w of the one-dimensional grid takes 256 items (each
// wavefront on its own) and one group of steps: w = group * ceil(n / 256) + block, so the
// workgroups that have work are the first ones dispatched and the surplus ones, which exit at
// once, come after all of them instead of taking wave slots in between.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_fog_noise_view(FogArgs a) {
    const uint32_t n = *a.item_count;
    const uint32_t blocks = (n + 255u) / 256u;
    if (blockIdx.x >= blocks * (uint32_t)kFogViewGroups) return;
    const uint32_t group = blockIdx.x / blocks, block = blockIdx.x % blocks;
    if ((block * 256u + (threadIdx.x & ~63u)) >= n) return;
    const uint32_t k = block * 256u + threadIdx.x;
    const bool live = k < n;
    const float4 ray = live ? a.item_ray[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const V3 origin = v3(ray.x, ray.y, ray.z);
    fog_noise_steps<COUNT>(a, live, origin, fog_item_dir(a.fc, origin), ray.w, kFogStepSize, kFogMaxSteps,
                           (int)group * kFogStepGroup, a.noise_view + k, &a.counters[5]);
}

// A workgroup takes kResolveChunks chunks of 256 consecutive items and reserves room in the lit
// list for all of them with ONE atomic: appends of a whole grid to one counter are serialised at
// about 12 ns each, which for one append per 256 items was a third of this kernel's time.  The
// list keeps the items in their order.
constexpr int kResolveChunks = 4;
template <bool COUNT>
__global__ __launch_bounds__(256) void k_fog_view_resolve(FogArgs a) {
    latency_bound_kernel();
    __shared__ uint32_t s_tot[kResolveChunks][4], s_base;
    const uint32_t n = *a.item_count;
    const uint32_t first = blockIdx.x * (256u * kResolveChunks);
    if (first >= n) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    V2 fm[kResolveChunks];
    float in_light[kResolveChunks];
    uint32_t before[kResolveChunks];  // lit items of the chunk in earlier lanes of this wavefront
#pragma unroll
    for (int c = 0; c < kResolveChunks; ++c) {
        const uint32_t k = first + (uint32_t)c * 256u + threadIdx.x;
        fm[c] = V2{0.0f, 0.0f};
        in_light[c] = 0.0f;
        if (k < n) {
            const float4 ray = a.item_ray[k];
            const V3 origin = v3(ray.x, ray.y, ray.z);
            uint32_t evals = 0;
            StoredNoise<kFogMaxSteps> nv;
            nv.fetch(a.noise_view + k, a.cap, kFogStepSize, ray.w);
            fm[c] = ray_march_stored(ray.w, kFogMaxSteps, kFogStepSize, kFogDensity, [&nv](int i) { return nv.v[i]; }, evals);
            const V3 fp = origin + fog_item_dir(a.fc, origin) * fm[c].y;
            ShadowMap sm{a.shadow, a.fc.S};
            in_light[c] = sample_direct_light(a.fc, sm, V4{fp.x, fp.y, fp.z, 1.0f});
            if (COUNT) atomicAdd(&a.counters[1], (unsigned long long)evals);
        }
        const unsigned long long lit_lanes = __ballot(in_light[c] > 0.0f);
        before[c] = (uint32_t)__builtin_popcountll(lit_lanes & ((1ull << lane) - 1ull));
        if (lane == 0) s_tot[c][wave] = (uint32_t)__builtin_popcountll(lit_lanes);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int c = 0; c < kResolveChunks; ++c)
            for (int w = 0; w < 4; ++w) total += s_tot[c][w];
        s_base = total ? atomicAdd(a.lit_count, total) : 0u;
    }
    __syncthreads();
    uint32_t j = s_base;
#pragma unroll
    for (int c = 0; c < kResolveChunks; ++c) {
        const uint32_t k = first + (uint32_t)c * 256u + threadIdx.x;
        uint32_t mine = j + before[c];
        for (int w = 0; w < 4; ++w) {
            if (w < wave) mine += s_tot[c][w];
            j += s_tot[c][w];
        }
        const bool lit = in_light[c] > 0.0f;
        if (lit) a.lit_item[mine] = k;
        if (k < n) a.item_res[k] = make_float4(fm[c].x, fm[c].y, in_light[c], u2f(lit ? mine : 0xFFFFFFFFu));
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_fog_noise_light(FogArgs a) {
    const uint32_t n = *a.lit_count;
    const uint32_t blocks = (n + 255u) / 256u;
    if (blockIdx.x >= blocks * (uint32_t)kFogLightGroups) return;
    const uint32_t group = blockIdx.x / blocks, block = blockIdx.x % blocks;
    if ((block * 256u + (threadIdx.x & ~63u)) >= n) return;
    const uint32_t j = block * 256u + threadIdx.x;
    const bool live = j < n;
    const uint32_t k = live ? a.lit_item[j] : 0u;
    const float4 ray = a.item_ray[k];
    const float4 res = a.item_res[k];
    const V3 origin = v3(ray.x, ray.y, ray.z);
    const LightRay lr = fog_light_ray(a.fc, origin + fog_item_dir(a.fc, origin) * res.y);
    fog_noise_steps<COUNT>(a, live, lr.origin, lr.dir, lr.max_dist, kFogLightStepSize, kFogLightMaxSteps,
                           (int)group * kFogStepGroup, a.noise_light + j, &a.counters[6]);
}

// Blends SrcAlpha / OneMinusSrcAlpha over the finished geometry colour (state.rs:501).
template <bool COUNT>
__global__ __launch_bounds__(256) void k_fog_shade_blend(FogArgs a) {
    latency_bound_kernel();
    __shared__ float s_tab[768];
    const uint32_t n = *a.item_count;
    if (blockIdx.x * 256u >= n) return;
    stage_color_tables(s_tab, a.tables);
    ColorTables ct{s_tab};
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const float4 ray = a.item_ray[k];
        const float4 res = a.item_res[k];
        const V3 origin = v3(ray.x, ray.y, ray.z);
        const V3 fp = origin + fog_item_dir(a.fc, origin) * res.y;
        const float in_light = res.z;
        float occlusion = 0.0f;
        if (in_light > 0.0f) {
            const LightRay lr = fog_light_ray(a.fc, fp);
            uint32_t evals = 0;
            StoredNoise<kFogLightMaxSteps> nl;
            nl.fetch(a.noise_light + f2u(res.w), a.cap, kFogLightStepSize, lr.max_dist);
            occlusion = ray_march_stored(lr.max_dist, kFogLightMaxSteps, kFogLightStepSize, kFogLightDensity,
                                         [&nl](int i) { return nl.v[i]; }, evals).x;
            if (COUNT) atomicAdd(&a.counters[1], (unsigned long long)evals);
        }
        const V4 src = fog_shade(a.fc, fp, res.x, in_light, occlusion);
        const uint32_t pi = a.item_pixel[k];
        a.color[pi] = blend_pack(ct, a.color[pi], src, true);
    }
}

// The marches of one work item in one piece (both marches of fog.wgsl, then the fog colour),
// one wavefront per workgroup: the schedule for frames that overlap on the device.  A launch
// of these long-running wavefronts thins out towards its end, which is where the short kernels
// of the neighbouring frames find their wave slots; the evenly sized units of the kernels above
// fill every slot to the end and suit a frame that has the device to itself.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_fog_march(FogArgs a) {
#ifdef WR_DEV_MARCH_VGPR
    // developer experiment: a register the kernel does not need, to hold it to fewer wavefronts per SIMD
    asm volatile("v_mov_b32 v" WR_DEV_MARCH_VGPR ", 0" ::: "v" WR_DEV_MARCH_VGPR);
#endif
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    if (k >= *a.item_count) return;
    const float4 ray = a.item_ray[k];
    ShadowMap sm{a.shadow, a.fc.S};
    uint32_t evals = 0;
    V4 src = fog_march_shade(a.fc, sm, &a.tables_noise, v3(ray.x, ray.y, ray.z), ray.w, evals);
    if (COUNT) atomicAdd(&a.counters[1], (unsigned long long)evals);
    a.item_res[k] = make_float4(src.x, src.y, src.z, src.w);
}

// Blends the fog colours k_fog_march left in item_res over the finished geometry colour
// (SrcAlpha / OneMinusSrcAlpha, state.rs:501).
__global__ __launch_bounds__(256) void k_fog_blend(FogArgs a) {
    latency_bound_kernel();
    __shared__ float s_tab[768];
    const uint32_t n = *a.item_count;
    if (blockIdx.x * 256u >= n) return;
    stage_color_tables(s_tab, a.tables);
    ColorTables ct{s_tab};
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        const uint32_t pi = a.item_pixel[k];
        const float4 c = a.item_res[k];
        a.color[pi] = blend_pack(ct, a.color[pi], V4{c.x, c.y, c.z, c.w}, true);
    }
}

// ------------------------------------------------------------------ bin pack / scatter
// Moves whole 64x64 colour bins between the frame and a packed [bin][64][64] buffer, for
// gathering a frame whose bins were rendered on different devices.
__global__ __launch_bounds__(256) void k_pack_bins(const uint32_t* color, uint32_t* packed, int W, int H, int bins_x,
                                                   int nbins, int rank, int world, int to_packed) {
    latency_bound_kernel();
    int j = blockIdx.x;                 // j-th bin owned by `rank`
    int b = rank + j * world;
    if (b >= nbins) return;
    int ox = (b % bins_x) * kBin, oy = (b / bins_x) * kBin;
    for (int i = threadIdx.x; i < kBin * kBin; i += 256) {
        int px = ox + (i & (kBin - 1)), py = oy + i / kBin;
        size_t po = (size_t)j * kBin * kBin + i;
        bool in = px < W && py < H;
        if (to_packed) packed[po] = in ? color[(size_t)py * W + px] : 0u;
        else if (in) const_cast<uint32_t*>(color)[(size_t)py * W + px] = packed[po];
    }
}

// The gathering rank's side of a bin gather: bin b of the frame belongs to rank b % world and is
// that rank's (b / world)-th bin; rank r's packed bins start at r * slot_bins in `gathered`.
// The root's own bins are in place already.
__global__ __launch_bounds__(256) void k_scatter_gathered(uint32_t* color, const uint32_t* gathered, int W, int H, int bins_x,
                                                          int nbins, int world, int root, int slot_bins) {
    latency_bound_kernel();
    const int b = blockIdx.x;
    if (b >= nbins) return;
    const int r = b % world, j = b / world;
    if (r == root) return;
    const uint32_t* src = gathered + ((size_t)r * slot_bins + j) * (kBin * kBin);
    const int ox = (b % bins_x) * kBin, oy = (b / bins_x) * kBin;
    for (int i = threadIdx.x; i < kBin * kBin; i += 256) {
        const int px = ox + (i & (kBin - 1)), py = oy + i / kBin;
        if (px < W && py < H) color[(size_t)py * W + px] = src[i];
    }
}

// Bgra8UnormSrgb target: the frame is composed as RGBA (every blend works per channel, so the
// order of the bytes does not enter any result) and the red and blue bytes change places once,
// at the end of the frame.
__global__ __launch_bounds__(256) void k_swap_red_blue(uint32_t* color, size_t n) {
    latency_bound_kernel();
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const uint32_t c = color[i];
        color[i] = (c & 0xFF00FF00u) | ((c & 0xFFu) << 16) | ((c >> 16) & 0xFFu);
    }
}

// Zeroes the per-frame counter blocks of one stream in a single launch.
struct ClearList { uint32_t* ptr[4]; uint32_t words[4]; };
__global__ __launch_bounds__(256) void k_clear_words(ClearList c) {
    latency_bound_kernel();
    for (int k = 0; k < 4; ++k)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < c.words[k]; i += gridDim.x * 256u) c.ptr[k][i] = 0u;
}

__global__ void k_fill_f32(float* p, size_t n, float v) {
    latency_bound_kernel();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace wr
