/*
 * wr_abi.h -- C ABI of libwr_hip.so, the MI355X (gfx950) implementation of the
 * per-frame render path of nullprop/wgpu-renderer.
 *
 * The reference has no FFI of its own: its frame orchestrator (src/core/state.rs)
 * talks to wgpu::Device / wgpu::Queue.  Every entry point below replaces one kind of
 * call the reference makes at that seam; the citation on each declaration names it.
 * Payload structs are byte-identical to the #[repr(C)] structs the reference uploads
 * with bytemuck, so a Rust host can pass `&self.camera_uniform as *const _` unchanged.
 *
 * Conventions
 *   - return 0 (WR_OK) on success, a negative WrStatus on failure; wr_last_error()
 *     gives a text for the last failure on that context.  Nothing throws or aborts.
 *   - the caller owns every input pointer; the library copies during the call.
 *   - one caller thread per context (the reference drives wgpu from the single winit
 *     event-loop thread, src/core/window.rs:55-69).  Calls take effect in call order, as
 *     wgpu queues execute in submission order.  A frame's kernels run on streams the library
 *     owns (three per frame in flight: geometry + shading, shadow passes, fog), forked from
 *     and -- with one frame in flight -- joined back into WrConfig.stream; see
 *     WrConfig.frames_in_flight for what the caller's stream may assume.
 *   - all matrices are column-major [[f32;4];4] exactly as cgmath/bytemuck lay them out.
 */
#ifndef WR_ABI_H
#define WR_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ payload structs */

/* src/core/model.rs:14-22 `ModelVertex` (56 B; offsets 0/12/20/32/44). */
typedef struct WrModelVertex {
    float position[3];
    float tex_coords[2];
    float normal[3];
    float tangent[3];
    float bitangent[3];
} WrModelVertex;

/* src/core/instance.rs:9-14 `InstanceRaw` (100 B): model 4x4 then normal 3x3. */
typedef struct WrInstanceRaw {
    float model[4][4];
    float normal[3][3];
} WrInstanceRaw;

/* src/core/camera.rs:104-112 `CameraUniform` (224 B). */
typedef struct WrCameraUniform {
    float view[4][4];
    float proj[4][4];
    float inv_view_proj[4][4];
    float position[4];
    float planes[4]; /* near, far, width, height */
} WrCameraUniform;

/* src/core/light.rs:11-18 `LightUniform` (416 B). */
typedef struct WrLightUniform {
    float position[3];
    uint32_t _padding;
    float color[4]; /* rgb + intensity */
    float matrices[6][4][4];
} WrLightUniform;

/* src/core/state.rs:21-28 `GlobalUniforms` (16 B).  light_matrix_index is ignored:
 * the library loops over the six cube faces itself (state.rs:633-675). */
typedef struct WrGlobalUniforms {
    float time;
    uint32_t light_matrix_index;
    uint32_t use_shadowmaps;
    uint32_t _padding;
} WrGlobalUniforms;

/* src/core/material.rs:13-18 `MaterialUniform` (16 B): metallic, roughness, 0, 0. */
typedef struct WrMaterialUniform {
    float factors[4];
} WrMaterialUniform;

/* ------------------------------------------------------------------ enums */

typedef enum WrStatus {
    WR_OK = 0,
    WR_ERR_INVALID_ARGUMENT = -1,
    WR_ERR_OUT_OF_MEMORY = -2,    /* window.rs:77 SurfaceError::OutOfMemory */
    WR_ERR_DEVICE = -3,           /* a HIP call failed (window.rs:73 SurfaceError::Lost) */
    WR_ERR_UNSUPPORTED = -4,
    WR_ERR_CAPACITY = -5,         /* an internal list overflowed its capacity */
    WR_ERR_NO_SCENE = -6
} WrStatus;

/* Which model a mesh / material / instance belongs to (state.rs:379-405: geom_model,
 * fog_model, light_model). */
typedef enum WrModel {
    WR_MODEL_GEOMETRY = 0,
    WR_MODEL_FOG_VOLUME = 1,
    WR_MODEL_LIGHT_MARKER = 2
} WrModel;

/* Which passes of State::render run.  The reference has no run-time switch; its wasm32
 * build is "no shadows, no fog" (state.rs:38,388,486,627,763) and native is all three. */
enum {
    WR_PASS_SHADOWS = 1u,     /* state.rs:633-675, six depth passes */
    WR_PASS_FOG = 2u,         /* state.rs:763-804 */
    WR_PASS_LIGHT_DEBUG = 4u, /* state.rs:728-759 */
    WR_PASS_ALL = 7u
};

/* Colour target formats (state.rs:106 takes caps.formats[0]: Bgra8UnormSrgb on most Vulkan
 * surfaces, Rgba8UnormSrgb on some).  Both hold the same values, bytes in the order named. */
enum { WR_FORMAT_RGBA8_UNORM_SRGB = 0, WR_FORMAT_BGRA8_UNORM_SRGB = 1 };

typedef struct WrConfig {
    uint32_t width;           /* state.rs:104-108 surface size */
    uint32_t height;
    uint32_t shadow_map_size; /* state.rs:18 SHADOW_MAP_SIZE (2048); 0 selects 2048 */
    uint32_t color_format;    /* state.rs:106 caps.formats[0]: WR_FORMAT_RGBA8_UNORM_SRGB or WR_FORMAT_BGRA8_UNORM_SRGB */
    int32_t device;           /* HIP device ordinal */
    /* Screen-bin sharding across contexts: this context produces only the 64x64-pixel
     * bins b (row-major bin index) with b % tile_world == tile_rank.  tile_world 0 or 1
     * renders the whole frame. */
    uint32_t tile_rank;
    uint32_t tile_world;
    /* hipStream_t the frames are ordered against, or NULL for a library-owned stream.  A frame
     * starts after the work queued on this stream before the wr_render_* call; its kernels run
     * on library-owned streams. */
    void* stream;
    /* Frames in flight (the reference's surface keeps more than one frame in flight as well,
     * state.rs:631,806).  0 or 1: a frame's work is complete, as seen from `stream`, before
     * anything queued on `stream` afterwards.  (Frames that follow each other on the library's
     * own stream with nothing queued in between are ordered buffer by buffer, not frame by
     * frame: a frame's vertex and triangle set-up may run while the previous frame's fog pass
     * ends.  No buffer is written before its last reader is done.)  2 to 4: consecutive frames rotate through that
     * many sets of frame buffers (about 1.5 GB each at 1080p) and overlap on the device; a
     * frame's results are complete after wr_sync (every wr_read_* and wr_get_stats syncs), and
     * read-back / wr_color_device_ptr refer to the most recently submitted frame.  More than
     * one frame in flight pays only when every stream has a HIP hardware queue of its own:
     * export GPU_MAX_HW_QUEUES=16 before the process's first HIP call (HIP's default of 4 makes
     * 2 to 4 frames in flight slower than 1). */
    uint32_t frames_in_flight;
    uint32_t reserved;
} WrConfig;

typedef struct WrContext WrContext;

/* Per-stage device times and per-frame counters.  Times are HIP-event milliseconds, averaged
 * over the (at most 128 most recent) frames rendered since wr_set_profiling(ctx, 1).  The
 * "counting mode" counters are those of the last wr_render_* call (summed over the poses of a
 * batch); a counted call waits on the device for the frames still in flight, so its counters
 * are its own.  The two *_bin_items counts are those of the most recently submitted frame.
 * The shadow pass runs on a library-owned side stream next to
 * the geometry pass, so ms_shadow and ms_geometry overlap, ms_shade includes any wait for the
 * shadow maps, and ms_total is the frame from first to last kernel, not the sum.  No
 * counterpart in the reference (timestamp_writes: None on every pass,
 * state.rs:660,713,749,788). */
typedef struct WrFrameStats {
    float ms_shadow;   /* shadow vertex + setup + binning + raster, six faces */
    float ms_geometry; /* geometry vertex + setup + binning + raster */
    float ms_shade;    /* deferred PBS shading incl. PCF and ordered blending */
    float ms_fog;      /* fog volume raster + marches, incl. any wait for the shadow maps */
    float ms_total;
    uint32_t profiled_frames;    /* frames the averages above cover */
    uint32_t geometry_bin_items; /* (triangle, 64x64 bin) pairs the geometry rasteriser processed */
    uint32_t shadow_bin_items;   /* same for the six shadow faces together */
    uint64_t fog_fragments;      /* fog fragments that passed the depth test (counting mode only) */
    uint64_t fog_noise_evals;    /* fog_noise evaluations the marches of fog.wgsl execute (counting mode only) */
    uint64_t shaded_fragments;   /* PBS fragments shaded (counting mode only) */
    uint64_t replay_pixels;      /* pixels that took the ordered-blend path (counting mode only) */
    uint64_t replay_slow_pixels; /* of those, pixels covered by too many fragments for the fast path */
    float ms_fog_march;          /* k_fog_noise_view alone: the noise of the view-march steps (part of ms_fog) */
    float ms_fog_light_march;    /* k_fog_noise_light alone: the noise of the light-march steps (part of ms_fog) */
    /* fog_noise evaluations the two noise kernels made (counting mode only): those of
     * fog_noise_evals plus the steps a march's density exit skipped (evaluated, never read) */
    uint64_t fog_noise_evals_view;
    uint64_t fog_noise_evals_light;
} WrFrameStats;

/* ------------------------------------------------------------------ context */

/* Replaces State::new's device/queue/surface/pipeline creation, state.rs:69-114,213-241,
 * 435-504 (the four RenderPass pipelines become fixed kernels). */
int wr_create(const WrConfig* config, WrContext** out_ctx);
void wr_destroy(WrContext* ctx);

/* Replaces State::resize, state.rs:575-587 (new colour target + geometry depth). */
int wr_resize(WrContext* ctx, uint32_t width, uint32_t height);

const char* wr_last_error(const WrContext* ctx);

/* ------------------------------------------------------------------ scene upload (load time) */

/* Replaces Texture::from_pixels -> queue.write_texture, texture.rs:74-116.  `srgb` != 0
 * is Rgba8UnormSrgb (base colour, resources.rs:62), 0 is Rgba8Unorm (resources.rs:89,116).
 * Texture ids are global to the context. */
int wr_upload_texture(WrContext* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height,
                      int srgb, uint32_t* out_texture);

/* Replaces Material::new, material.rs:21-87.  Material ids count from 0 per model in
 * call order, like `materials.push` in resources.rs:121. */
int wr_upload_material(WrContext* ctx, uint32_t model, uint32_t tex_base_color,
                       uint32_t tex_normal, uint32_t tex_metallic_roughness,
                       float metallic_factor, float roughness_factor, uint32_t* out_material);

/* Replaces the vertex/index create_buffer_init of Mesh::from_gltf, mesh.rs:91-108.
 * Call order is draw order (model.rs:147-156); it defines global triangle ids:
 * id = (triangles of earlier meshes of the model) + local triangle index. */
int wr_upload_mesh(WrContext* ctx, uint32_t model, const WrModelVertex* vertices,
                   uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                   uint32_t material);

/* Replaces the instance create_buffer_init calls, state.rs:413-433.  One instance per
 * model (the reference draws 0..1 of each). */
int wr_set_instances(WrContext* ctx, uint32_t model, const WrInstanceRaw* instances,
                     uint32_t count);

/* Drops every mesh, material and texture (no counterpart; the reference loads once). */
int wr_clear_scene(WrContext* ctx);

/* ------------------------------------------------------------------ per frame */

/* Replaces the three queue.write_buffer calls of State::update (state.rs:603-607,619-623,
 * 635-639) plus all of State::render (state.rs:630-809).  Asynchronous: the frame is queued
 * behind what WrConfig.stream holds at the time of the call and runs on the library's own
 * streams.  `passes` is a mask of WR_PASS_*.  A frame without WR_PASS_SHADOWS samples cleared
 * shadow maps (depth 1.0: nothing is in shadow) if globals->use_shadowmaps is set; the
 * reference's configuration without shadow passes sets it to 0 (state.rs:627). */
int wr_render_frame(WrContext* ctx, const WrCameraUniform* camera, const WrLightUniform* light,
                    const WrGlobalUniforms* globals, uint32_t passes);

/* Frame-parallel throughput mode: renders `count` camera poses under one light and one
 * time value.  The six shadow passes run once and are reused unless
 * `shadows_every_frame` != 0 (then every pose repeats them, as consecutive State::render
 * calls would).  If `device_frames` is not NULL it must be device memory for
 * count*width*height*4 bytes and receives every finished frame; otherwise only the last
 * frame stays readable.  count == 0 renders nothing and leaves the readable frame as it is. */
int wr_render_batch(WrContext* ctx, const WrCameraUniform* cameras, uint32_t count,
                    const WrLightUniform* light, const WrGlobalUniforms* globals,
                    uint32_t passes, int shadows_every_frame, void* device_frames);

/* wr_render_frame for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, hosts that must have every allocation behind them before a
 * region that may not allocate begins).  No counterpart in the reference.  Renders the same
 * frame as wr_render_frame, bit for bit, with every work buffer of the frame (shadow maps, bin
 * lists, visibility, ordered-blend lists, fog items, noise tables) inside one file-scope device
 * array of 96 MiB that each GPU using the library holds until the process ends; the finished
 * RGBA8 frame goes to `device_color` (device memory, width*height*4 bytes, brought by the
 * caller).  Calls no hipMalloc / hipFree.  The call itself cannot be recorded into a HIP graph:
 * it forks onto the library's streams and waits on events of earlier, unrecorded frames.  The
 * scene must already be resident: the first wr_render_frame after an upload or resize makes it
 * so, otherwise WR_ERR_NO_SCENE.  A target and scene whose work buffers do not fit the array
 * (roughly: beyond 320x180 pixels with 256^2 shadow faces and 16k triangles; the error text
 * gives the exact need) return WR_ERR_CAPACITY; bin-sharded contexts return
 * WR_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's array, from any context and any
 * host thread: the library orders them on the device, each behind the one submitted before it
 * (an event per GPU; submission is under a mutex), so they never overlap there; anything else
 * a caller runs next to them is untouched.  Results are complete, as seen from the context's
 * stream, when the call's work has run (wr_sync); wr_read_* keep referring to
 * wr_render_frame's buffers; a list overflow of a static frame is reported by wr_sync as such. */
int wr_render_frame_static(WrContext* ctx, const WrCameraUniform* camera, const WrLightUniform* light,
                           const WrGlobalUniforms* globals, uint32_t passes, void* device_color);

/* Blocks until the context's stream is idle (surface_texture.present(), state.rs:806,
 * is the reference's end-of-frame point).  This is also where a frame that overflowed one of
 * the library's lists is reported (WR_ERR_CAPACITY; the text names the frame by its number in
 * submission order: each set of frame buffers has its own error word). */
int wr_sync(WrContext* ctx);

/* ------------------------------------------------------------------ results */

/* Synchronous copies to host memory.  Colour is RGBA8 or BGRA8 (WrConfig.color_format;
 * sRGB-encoded), row-major, top row first: the surface texture of state.rs:678-681.  Depth is the geometry depth texture
 * (state.rs:562-573).  Visibility is, per pixel, (triangle id << 3 | clip-fan index) of
 * the fragment that owns the depth value, 0xFFFFFFFF where nothing was drawn; ids of the
 * light marker continue after the geometry model's.  Shadow is the 6-layer depth array
 * (state.rs:215-224), layer-major. */
int wr_read_color(WrContext* ctx, uint8_t* dst, size_t bytes);
int wr_read_depth(WrContext* ctx, float* dst, size_t bytes);
int wr_read_visibility(WrContext* ctx, uint32_t* dst, size_t bytes);
int wr_read_shadow_map(WrContext* ctx, float* dst, size_t bytes);

/* Device address of the RGBA8 colour target, for collectives that gather tiles without a
 * host round trip.  Valid until wr_resize / wr_destroy. */
void* wr_color_device_ptr(WrContext* ctx);

/* Copies the bins this context owns (see WrConfig.tile_rank) into `device_dst`, packed in
 * ascending bin order as 64x64x4-byte blocks; returns the number of bins through
 * `out_bins`.  wr_scatter_bins is the inverse on the gathering side: it writes `bins`
 * packed blocks that belong to `rank` of `world` into this context's colour target. */
int wr_pack_owned_bins(WrContext* ctx, void* device_dst, uint32_t* out_bins);
int wr_scatter_bins(WrContext* ctx, const void* device_src, uint32_t rank, uint32_t world);

/* ---- gathering a bin-sharded frame (one 3840x2160 frame, its 64x64 bins dealt
 * round-robin over the ranks).  No counterpart in the reference, which is single-GPU; the
 * frame arriving complete on one device stands for its `surface_texture.present()`
 * (state.rs:806).
 *
 * wr_gather_layout: rank `rank` of `world` owns bins rank, rank + world, ... of the row-major
 * bin grid; `*owned` of them.  Every rank's packed bins occupy a slot of `*slot_bins` bins
 * (ceil(bins / world)) in the root's receive area, rank r's slot at bin offset r * *slot_bins.
 * Pure arithmetic: needs no device. */
void wr_gather_layout(uint32_t width, uint32_t height, uint32_t world, uint32_t rank,
                      uint32_t* owned, uint32_t* slot_bins);

/* How packed bins travel between ranks.  `send` / `recv` move `bytes` of device memory to /
 * from rank `peer`, ordered on the HIP stream `stream`; a group brackets the transfers of one
 * gather like ncclGroupStart / ncclGroupEnd.  Each returns 0 on success.  The library calls
 * them synchronously, on the caller's thread, inside wr_gather_bins_via only.  wr_gather_bins
 * uses RCCL; tests bring a host-side stand-in. */
typedef struct WrTransport {
    void* user;
    int (*group_start)(void* user);
    int (*send)(void* user, const void* device_buf, size_t bytes, int peer, void* stream);
    int (*recv)(void* user, void* device_buf, size_t bytes, int peer, void* stream);
    int (*group_end)(void* user);
} WrTransport;

/* Collects the most recent frame of a bin-sharded set of contexts (WrConfig.tile_world > 1,
 * one context per rank) on rank `root`: every other rank packs its bins and sends them, the
 * root receives into its slots and writes the bins into its own colour target, which then
 * holds the whole frame.  Everything is queued on the context's stream behind the frame:
 * no host synchronisation between rendering, packing and the transfers.  Every rank of the
 * frame must make the call. */
int wr_gather_bins_via(WrContext* ctx, const WrTransport* transport, uint32_t root);
/* The same over RCCL: `nccl_comm` is an ncclComm_t of `tile_world` ranks in which this
 * context's rank is `tile_rank` (grouped ncclSend / ncclRecv straight to the root: one hop
 * on the root's own xGMI link to each peer). */
int wr_gather_bins(WrContext* ctx, void* nccl_comm, uint32_t root);
/* RCCL communicator helpers for hosts that do not bind RCCL themselves (the library loads
 * librccl.so.1 on first use): rank 0 makes an id, every rank gets the 128 bytes by the host's
 * own means and creates its communicator on its context's device. */
int wr_comm_unique_id(uint8_t id[128]);
int wr_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, int32_t device, void** out_comm);
int wr_comm_destroy(void* comm);
/* The transfers wr_gather_bins uses, for hosts that compose their own exchanges from them:
 * fills `out` with grouped ncclSend / ncclRecv on `comm` (bytes as ncclUint8).  Valid until
 * wr_comm_destroy(comm). */
int wr_comm_transport(void* comm, WrTransport* out);
/* File name of the RCCL library the loader bound (a process can hold more than one build under
 * the same soname, e.g. a framework's bundled copy next to /opt/rocm's): diagnostics. */
int wr_comm_library_path(char* path, size_t cap);

/* enabled: 0 off; 1 records HIP events around every stage (a dozen per frame: a few per cent of a
 * frame that has the device to itself); 2 only the pair around the dominant kernel (the view
 * noise or the march), which fills ms_fog_march and leaves the other times at zero; with one
 * frame in flight, where each event costs the frame about 1 %, level 2 records the pair for every
 * fourth frame (WrFrameStats.profiled_frames counts the recorded ones). */
int wr_set_profiling(WrContext* ctx, int enabled);
/* enabled != 0 makes the next frames also count fragments / march steps with atomics
 * (slower, and a counted call does not overlap the frames before it; never enable it inside
 * a timed region). */
int wr_set_counting(WrContext* ctx, int enabled);
int wr_get_stats(WrContext* ctx, WrFrameStats* out);

/* ------------------------------------------------------------------ GLB import (host only)
 *
 * Replaces load_model_gltf (resources.rs:17-136) and Mesh::from_gltf (mesh.rs:13-112), with
 * their quirks: one mesh per primitive in document order, node transforms ignored,
 * tangent = TANGENT.xyz * w, bitangent = normal x tangent, NaN tangents without TANGENT,
 * missing texture slots fall back to image 0, RGB padded to RGBA, base colour sRGB and the
 * other two maps linear, primitive.material defaults to 0.  Embedded 8-bit PNG and JPEG
 * (baseline, extended sequential, progressive; three components) images are decoded; 16-bit,
 * grey and CMYK images return WR_ERR_UNSUPPORTED (the reference panics on 16-bit and grey
 * ones, resources.rs:51,138-151).  Parsing touches no device. */
typedef struct WrGlb WrGlb;
int wr_glb_parse(const void* data, size_t bytes, WrGlb** out, char* err, size_t err_cap);
int wr_glb_load_file(const char* path, WrGlb** out, char* err, size_t err_cap);
void wr_glb_free(WrGlb* glb);
void wr_glb_counts(const WrGlb* glb, uint32_t* meshes, uint32_t* materials, uint32_t* textures);
/* Pointers stay valid until wr_glb_free.  Material texture ids index wr_glb_texture. */
int wr_glb_mesh(const WrGlb* glb, uint32_t i, const WrModelVertex** vertices, uint32_t* vertex_count,
                const uint32_t** indices, uint32_t* index_count, uint32_t* material);
int wr_glb_texture(const WrGlb* glb, uint32_t i, const uint8_t** rgba8, uint32_t* width, uint32_t* height, int* srgb);
int wr_glb_material(const WrGlb* glb, uint32_t i, uint32_t* base, uint32_t* normal, uint32_t* mr,
                    float* metallic, float* roughness);
/* The upload sequence of load_model_gltf: textures, materials, then meshes in draw order. */
int wr_upload_glb(WrContext* ctx, uint32_t model, const WrGlb* glb);

/* ------------------------------------------------------------------ host scene math
 * The Rust host keeps these (src/core/camera.rs, light.rs, instance.rs); they are
 * provided so that a C/C++/Python host produces the same uniform bytes. */

/* camera.rs:56-76 get_view_matrix/get_vecs + camera.rs:25-33 + camera.rs:125-134. */
void wr_camera_uniform_update(WrCameraUniform* out, const float position[3], float pitch_deg,
                              float yaw_deg, float fovy_deg, uint32_t width, uint32_t height);
/* light.rs:32-42 update_matrices: six perspective(90 deg,1,near,far) * look_to_rh. */
void wr_light_update_matrices(WrLightUniform* light);
/* state.rs:609-617 light animation + light.rs:32; color[3] (intensity) is left as is. */
void wr_light_animate(WrLightUniform* light, float time_seconds);
/* instance.rs:16-25 Instance::to_raw; rotation is a quaternion (w, x, y, z). */
void wr_instance_to_raw(WrInstanceRaw* out, const float position[3], const float rotation_wxyz[4],
                        const float scale[3]);
/* camera.rs:78-101 Camera::update (fly controls).  pose = {x,y,z,pitch,yaw};
 * input = {deltax, deltay, forward-backward, right-left, up-down}. */
void wr_camera_fly_update(float pose[5], const float input[5], float speed, float sensitivity,
                          float dt_seconds);

#ifdef __cplusplus
}
#endif
#endif /* WR_ABI_H */
