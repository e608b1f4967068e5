//! `wr_sys` -- raw bindings to libwr_hip.so (include/wr_abi.h), the MI355X implementation of
//! the frame that `State::render` draws (src/core/state.rs:630-809).
//!
//! Drop this file into the reference as `src/core/wr_sys.rs` and add `pub mod wr_sys;` to
//! src/core/mod.rs.  Every item of the header has its counterpart here, in the header's order:
//! tests/test_rust_binding.py checks names, argument counts, argument types and struct fields
//! of this file against the header, so the two cannot drift apart unnoticed.
//!
//! The payload structs (`WrModelVertex` ...) have the layout of the reference's own
//! `#[repr(C)]` + `bytemuck::Pod` structs (model.rs:14-22, instance.rs:9-14, camera.rs:104-112,
//! light.rs:11-18, state.rs:21-28, material.rs:13-18), so a host passes
//! `&self.camera_uniform as *const CameraUniform as *const WrCameraUniform`.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

// ------------------------------------------------------------------ payload structs

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrModelVertex {
    pub position: [f32; 3],
    pub tex_coords: [f32; 2],
    pub normal: [f32; 3],
    pub tangent: [f32; 3],
    pub bitangent: [f32; 3],
}

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrInstanceRaw {
    pub model: [[f32; 4]; 4],
    pub normal: [[f32; 3]; 3],
}

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrCameraUniform {
    pub view: [[f32; 4]; 4],
    pub proj: [[f32; 4]; 4],
    pub inv_view_proj: [[f32; 4]; 4],
    pub position: [f32; 4],
    pub planes: [f32; 4],
}

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrLightUniform {
    pub position: [f32; 3],
    pub _padding: u32,
    pub color: [f32; 4],
    pub matrices: [[[f32; 4]; 4]; 6],
}

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrGlobalUniforms {
    pub time: f32,
    pub light_matrix_index: u32,
    pub use_shadowmaps: u32,
    pub _padding: u32,
}

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrMaterialUniform {
    pub factors: [f32; 4],
}

// ------------------------------------------------------------------ enums

pub const WR_OK: c_int = 0;
pub const WR_ERR_INVALID_ARGUMENT: c_int = -1;
pub const WR_ERR_OUT_OF_MEMORY: c_int = -2;
pub const WR_ERR_DEVICE: c_int = -3;
pub const WR_ERR_UNSUPPORTED: c_int = -4;
pub const WR_ERR_CAPACITY: c_int = -5;
pub const WR_ERR_NO_SCENE: c_int = -6;

pub const WR_MODEL_GEOMETRY: u32 = 0;
pub const WR_MODEL_FOG_VOLUME: u32 = 1;
pub const WR_MODEL_LIGHT_MARKER: u32 = 2;

pub const WR_PASS_SHADOWS: u32 = 1;
pub const WR_PASS_FOG: u32 = 2;
pub const WR_PASS_LIGHT_DEBUG: u32 = 4;
pub const WR_PASS_ALL: u32 = 7;

pub const WR_FORMAT_RGBA8_UNORM_SRGB: u32 = 0;
pub const WR_FORMAT_BGRA8_UNORM_SRGB: u32 = 1;

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct WrConfig {
    pub width: u32,
    pub height: u32,
    pub shadow_map_size: u32,
    pub color_format: u32,
    pub device: i32,
    pub tile_rank: u32,
    pub tile_world: u32,
    pub stream: *mut c_void,
    pub frames_in_flight: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct WrContext {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Copy, Clone, Debug, Default)]
pub struct WrFrameStats {
    pub ms_shadow: f32,
    pub ms_geometry: f32,
    pub ms_shade: f32,
    pub ms_fog: f32,
    pub ms_total: f32,
    pub profiled_frames: u32,
    pub geometry_bin_items: u32,
    pub shadow_bin_items: u32,
    pub fog_fragments: u64,
    pub fog_noise_evals: u64,
    pub shaded_fragments: u64,
    pub replay_pixels: u64,
    pub replay_slow_pixels: u64,
    pub ms_fog_march: f32,
    pub ms_fog_light_march: f32,
    pub fog_noise_evals_view: u64,
    pub fog_noise_evals_light: u64,
}

/// How packed bins travel between ranks in `wr_gather_bins_via`.
#[repr(C)]
#[derive(Copy, Clone)]
pub struct WrTransport {
    pub user: *mut c_void,
    pub group_start: Option<unsafe extern "C" fn(user: *mut c_void) -> c_int>,
    pub send: Option<unsafe extern "C" fn(user: *mut c_void, device_buf: *const c_void, bytes: usize, peer: c_int, stream: *mut c_void) -> c_int>,
    pub recv: Option<unsafe extern "C" fn(user: *mut c_void, device_buf: *mut c_void, bytes: usize, peer: c_int, stream: *mut c_void) -> c_int>,
    pub group_end: Option<unsafe extern "C" fn(user: *mut c_void) -> c_int>,
}

#[repr(C)]
pub struct WrGlb {
    _private: [u8; 0],
}

#[link(name = "wr_hip")]
extern "C" {
    // ---- context (State::new, state.rs:69-114,213-241,435-504; State::resize, state.rs:575-587)
    pub fn wr_create(config: *const WrConfig, out_ctx: *mut *mut WrContext) -> c_int;
    pub fn wr_destroy(ctx: *mut WrContext);
    pub fn wr_resize(ctx: *mut WrContext, width: u32, height: u32) -> c_int;
    pub fn wr_last_error(ctx: *const WrContext) -> *const c_char;

    // ---- scene upload (texture.rs:74-116, material.rs:21-87, mesh.rs:91-108, state.rs:413-433)
    pub fn wr_upload_texture(ctx: *mut WrContext, rgba8: *const u8, width: u32, height: u32, srgb: c_int, out_texture: *mut u32) -> c_int;
    pub fn wr_upload_material(ctx: *mut WrContext, model: u32, tex_base_color: u32, tex_normal: u32, tex_metallic_roughness: u32, metallic_factor: f32, roughness_factor: f32, out_material: *mut u32) -> c_int;
    pub fn wr_upload_mesh(ctx: *mut WrContext, model: u32, vertices: *const WrModelVertex, vertex_count: u32, indices: *const u32, index_count: u32, material: u32) -> c_int;
    pub fn wr_set_instances(ctx: *mut WrContext, model: u32, instances: *const WrInstanceRaw, count: u32) -> c_int;
    pub fn wr_clear_scene(ctx: *mut WrContext) -> c_int;

    // ---- per frame (State::update's write_buffer calls + State::render, state.rs:603-639,630-809)
    pub fn wr_render_frame(ctx: *mut WrContext, camera: *const WrCameraUniform, light: *const WrLightUniform, globals: *const WrGlobalUniforms, passes: u32) -> c_int;
    pub fn wr_render_batch(ctx: *mut WrContext, cameras: *const WrCameraUniform, count: u32, light: *const WrLightUniform, globals: *const WrGlobalUniforms, passes: u32, shadows_every_frame: c_int, device_frames: *mut c_void) -> c_int;
    pub fn wr_render_frame_static(ctx: *mut WrContext, camera: *const WrCameraUniform, light: *const WrLightUniform, globals: *const WrGlobalUniforms, passes: u32, device_color: *mut c_void) -> c_int;
    pub fn wr_sync(ctx: *mut WrContext) -> c_int;

    // ---- results (surface texture state.rs:678-681, depth texture state.rs:562-573, shadow array state.rs:215-224)
    pub fn wr_read_color(ctx: *mut WrContext, dst: *mut u8, bytes: usize) -> c_int;
    pub fn wr_read_depth(ctx: *mut WrContext, dst: *mut f32, bytes: usize) -> c_int;
    pub fn wr_read_visibility(ctx: *mut WrContext, dst: *mut u32, bytes: usize) -> c_int;
    pub fn wr_read_shadow_map(ctx: *mut WrContext, dst: *mut f32, bytes: usize) -> c_int;
    pub fn wr_color_device_ptr(ctx: *mut WrContext) -> *mut c_void;
    pub fn wr_pack_owned_bins(ctx: *mut WrContext, device_dst: *mut c_void, out_bins: *mut u32) -> c_int;
    pub fn wr_scatter_bins(ctx: *mut WrContext, device_src: *const c_void, rank: u32, world: u32) -> c_int;

    // ---- gathering a bin-sharded frame (several GPUs; stands for present(), state.rs:806)
    pub fn wr_gather_layout(width: u32, height: u32, world: u32, rank: u32, owned: *mut u32, slot_bins: *mut u32);
    pub fn wr_gather_bins_via(ctx: *mut WrContext, transport: *const WrTransport, root: u32) -> c_int;
    pub fn wr_gather_bins(ctx: *mut WrContext, nccl_comm: *mut c_void, root: u32) -> c_int;
    pub fn wr_comm_unique_id(id: *mut u8) -> c_int;
    pub fn wr_comm_create(id: *const u8, rank: u32, world: u32, device: i32, out_comm: *mut *mut c_void) -> c_int;
    pub fn wr_comm_destroy(comm: *mut c_void) -> c_int;
    pub fn wr_comm_transport(comm: *mut c_void, out: *mut WrTransport) -> c_int;
    pub fn wr_comm_library_path(path: *mut c_char, cap: usize) -> c_int;

    // ---- timing and counters
    pub fn wr_set_profiling(ctx: *mut WrContext, enabled: c_int) -> c_int;
    pub fn wr_set_counting(ctx: *mut WrContext, enabled: c_int) -> c_int;
    pub fn wr_get_stats(ctx: *mut WrContext, out: *mut WrFrameStats) -> c_int;

    // ---- GLB import (load_model_gltf, resources.rs:17-136; Mesh::from_gltf, mesh.rs:13-112)
    pub fn wr_glb_parse(data: *const c_void, bytes: usize, out: *mut *mut WrGlb, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn wr_glb_load_file(path: *const c_char, out: *mut *mut WrGlb, err: *mut c_char, err_cap: usize) -> c_int;
    pub fn wr_glb_free(glb: *mut WrGlb);
    pub fn wr_glb_counts(glb: *const WrGlb, meshes: *mut u32, materials: *mut u32, textures: *mut u32);
    pub fn wr_glb_mesh(glb: *const WrGlb, i: u32, vertices: *mut *const WrModelVertex, vertex_count: *mut u32, indices: *mut *const u32, index_count: *mut u32, material: *mut u32) -> c_int;
    pub fn wr_glb_texture(glb: *const WrGlb, i: u32, rgba8: *mut *const u8, width: *mut u32, height: *mut u32, srgb: *mut c_int) -> c_int;
    pub fn wr_glb_material(glb: *const WrGlb, i: u32, base: *mut u32, normal: *mut u32, mr: *mut u32, metallic: *mut f32, roughness: *mut f32) -> c_int;
    pub fn wr_upload_glb(ctx: *mut WrContext, model: u32, glb: *const WrGlb) -> c_int;

    // ---- host scene math (camera.rs, light.rs, instance.rs; a Rust host keeps its own)
    pub fn wr_camera_uniform_update(out: *mut WrCameraUniform, position: *const f32, pitch_deg: f32, yaw_deg: f32, fovy_deg: f32, width: u32, height: u32);
    pub fn wr_light_update_matrices(light: *mut WrLightUniform);
    pub fn wr_light_animate(light: *mut WrLightUniform, time_seconds: f32);
    pub fn wr_instance_to_raw(out: *mut WrInstanceRaw, position: *const f32, rotation_wxyz: *const f32, scale: *const f32);
    pub fn wr_camera_fly_update(pose: *mut f32, input: *const f32, speed: f32, sensitivity: f32, dt_seconds: f32);
}

/// Owning handle used by the `State` patch (bindings/rust/state_patch.md).
pub struct Renderer {
    pub ctx: *mut WrContext,
}

impl Renderer {
    pub fn new(config: &WrConfig) -> Result<Self, c_int> {
        let mut ctx: *mut WrContext = std::ptr::null_mut();
        let rc = unsafe { wr_create(config, &mut ctx) };
        if rc == WR_OK { Ok(Self { ctx }) } else { Err(rc) }
    }

    pub fn check(&self, rc: c_int) -> Result<(), String> {
        if rc == WR_OK {
            return Ok(());
        }
        let text = unsafe { std::ffi::CStr::from_ptr(wr_last_error(self.ctx)) };
        Err(format!("libwr_hip: status {rc}: {}", text.to_string_lossy()))
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { wr_destroy(self.ctx) }
    }
}
