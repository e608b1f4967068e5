"""ctypes binding of libwr_hip.so (include/wr_abi.h).

This is the Python stand-in for the `extern "C"` block a Rust host would declare; the
payload structs mirror the reference's #[repr(C)] structs byte for byte
(src/core/model.rs:14-22, instance.rs:9-14, camera.rs:104-112, light.rs:11-18,
state.rs:21-28).  There is no fallback: if the shared library is missing, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WR_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libwr_hip.so")

WR_OK = 0
WR_ERR_INVALID_ARGUMENT = -1
WR_ERR_OUT_OF_MEMORY = -2
WR_ERR_DEVICE = -3
WR_ERR_UNSUPPORTED = -4
FORMAT_RGBA8_UNORM_SRGB, FORMAT_BGRA8_UNORM_SRGB = 0, 1
WR_ERR_CAPACITY = -5
WR_ERR_NO_SCENE = -6

MODEL_GEOMETRY, MODEL_FOG_VOLUME, MODEL_LIGHT_MARKER = 0, 1, 2
PASS_SHADOWS, PASS_FOG, PASS_LIGHT_DEBUG, PASS_ALL = 1, 2, 4, 7

VERTEX_FLOATS = 14  # position3 uv2 normal3 tangent3 bitangent3 = 56 bytes


class InstanceRaw(C.Structure):
    _fields_ = [("model", (C.c_float * 4) * 4), ("normal", (C.c_float * 3) * 3)]


class CameraUniform(C.Structure):
    _fields_ = [
        ("view", (C.c_float * 4) * 4),
        ("proj", (C.c_float * 4) * 4),
        ("inv_view_proj", (C.c_float * 4) * 4),
        ("position", C.c_float * 4),
        ("planes", C.c_float * 4),
    ]


class LightUniform(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("_padding", C.c_uint32),
        ("color", C.c_float * 4),
        ("matrices", ((C.c_float * 4) * 4) * 6),
    ]


class GlobalUniforms(C.Structure):
    _fields_ = [
        ("time", C.c_float),
        ("light_matrix_index", C.c_uint32),
        ("use_shadowmaps", C.c_uint32),
        ("_padding", C.c_uint32),
    ]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("shadow_map_size", C.c_uint32),
        ("color_format", C.c_uint32),
        ("device", C.c_int32),
        ("tile_rank", C.c_uint32),
        ("tile_world", C.c_uint32),
        ("stream", C.c_void_p),
        ("frames_in_flight", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("ms_shadow", C.c_float),
        ("ms_geometry", C.c_float),
        ("ms_shade", C.c_float),
        ("ms_fog", C.c_float),
        ("ms_total", C.c_float),
        ("profiled_frames", C.c_uint32),
        ("geometry_bin_items", C.c_uint32),
        ("shadow_bin_items", C.c_uint32),
        ("fog_fragments", C.c_uint64),
        ("fog_noise_evals", C.c_uint64),
        ("shaded_fragments", C.c_uint64),
        ("replay_pixels", C.c_uint64),
        ("replay_slow_pixels", C.c_uint64),
        ("ms_fog_march", C.c_float),
        ("ms_fog_light_march", C.c_float),
        ("fog_noise_evals_view", C.c_uint64),
        ("fog_noise_evals_light", C.c_uint64),
    ]


assert C.sizeof(InstanceRaw) == 100 and C.sizeof(CameraUniform) == 224
assert C.sizeof(LightUniform) == 416 and C.sizeof(GlobalUniforms) == 16

class Transport(C.Structure):
    """WrTransport: how packed bins travel between ranks (wr_gather_bins_via)."""
    GROUP = C.CFUNCTYPE(C.c_int, C.c_void_p)
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("group_start", GROUP), ("send", SEND), ("recv", RECV), ("group_end", GROUP)]


# every symbol include/wr_abi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "wr_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "wr_destroy": (None, [_P]),
    "wr_resize": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "wr_last_error": (C.c_char_p, [_P]),
    "wr_upload_texture": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]),
    "wr_upload_material": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_uint32)]),
    "wr_upload_mesh": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32]),
    "wr_set_instances": (C.c_int, [_P, C.c_uint32, C.POINTER(InstanceRaw), C.c_uint32]),
    "wr_clear_scene": (C.c_int, [_P]),
    "wr_render_frame": (C.c_int, [_P, C.POINTER(CameraUniform), C.POINTER(LightUniform), C.POINTER(GlobalUniforms), C.c_uint32]),
    "wr_render_frame_static": (C.c_int, [_P, C.POINTER(CameraUniform), C.POINTER(LightUniform), C.POINTER(GlobalUniforms), C.c_uint32, _P]),
    "wr_render_batch": (C.c_int, [_P, C.POINTER(CameraUniform), C.c_uint32, C.POINTER(LightUniform), C.POINTER(GlobalUniforms), C.c_uint32, C.c_int, _P]),
    "wr_sync": (C.c_int, [_P]),
    "wr_read_color": (C.c_int, [_P, _P, C.c_size_t]),
    "wr_read_depth": (C.c_int, [_P, _P, C.c_size_t]),
    "wr_read_visibility": (C.c_int, [_P, _P, C.c_size_t]),
    "wr_read_shadow_map": (C.c_int, [_P, _P, C.c_size_t]),
    "wr_color_device_ptr": (_P, [_P]),
    "wr_pack_owned_bins": (C.c_int, [_P, _P, C.POINTER(C.c_uint32)]),
    "wr_scatter_bins": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "wr_gather_layout": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "wr_gather_bins_via": (C.c_int, [_P, _P, C.c_uint32]),
    "wr_gather_bins": (C.c_int, [_P, _P, C.c_uint32]),
    "wr_comm_unique_id": (C.c_int, [_P]),
    "wr_comm_create": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(_P)]),
    "wr_comm_destroy": (C.c_int, [_P]),
    "wr_comm_transport": (C.c_int, [_P, _P]),
    "wr_comm_library_path": (C.c_int, [C.c_char_p, C.c_size_t]),
    "wr_glb_parse": (C.c_int, [_P, C.c_size_t, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "wr_glb_load_file": (C.c_int, [C.c_char_p, C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "wr_glb_free": (None, [_P]),
    "wr_glb_counts": (None, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "wr_glb_mesh": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "wr_glb_texture": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "wr_glb_material": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wr_upload_glb": (C.c_int, [_P, C.c_uint32, _P]),
    "wr_set_profiling": (C.c_int, [_P, C.c_int]),
    "wr_set_counting": (C.c_int, [_P, C.c_int]),
    "wr_get_stats": (C.c_int, [_P, C.POINTER(FrameStats)]),
    "wr_camera_uniform_update": (None, [C.POINTER(CameraUniform), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32]),
    "wr_light_update_matrices": (None, [C.POINTER(LightUniform)]),
    "wr_light_animate": (None, [C.POINTER(LightUniform), C.c_float]),
    "wr_instance_to_raw": (None, [C.POINTER(InstanceRaw), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wr_camera_fly_update": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float]),
}

_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads libwr_hip.so and types every entry point.  Raises if the library or a symbol
    is missing: there is no CPU substitute for the HIP path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `make -C wgpu-renderer_amd/csrc` (hipcc, gfx950)")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class WrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"wr error {code}: {msg}")
        self.code = code


def _fptr(a: Sequence[float], n: int):
    return (C.c_float * n)(*[float(x) for x in a])


# ---------------------------------------------------------------- host scene math wrappers
def camera_uniform(position, pitch_deg, yaw_deg, fovy_deg, width, height) -> CameraUniform:
    """camera.rs:56-76,25-33,125-134 through wr_camera_uniform_update."""
    cu = CameraUniform()
    load_library().wr_camera_uniform_update(C.byref(cu), _fptr(position, 3), pitch_deg, yaw_deg, fovy_deg, width, height)
    return cu


def light_uniform(position=(0.0, 0.0, 0.0), color=(1.0, 1.0, 1.0, 250000.0)) -> LightUniform:
    """light.rs:21-30 LightUniform::new."""
    lu = LightUniform()
    for k in range(3):
        lu.position[k] = position[k]
    for k in range(4):
        lu.color[k] = color[k]
    load_library().wr_light_update_matrices(C.byref(lu))
    return lu


def light_at_time(t: float, intensity: float = 250000.0) -> LightUniform:
    """state.rs:134 initial light + state.rs:609-617 animation at time t."""
    lu = light_uniform((0.0, 0.0, 0.0), (1.0, 1.0, 1.0, intensity))
    load_library().wr_light_animate(C.byref(lu), float(t))
    return lu


def instance_raw(position, rotation_wxyz=(1.0, 0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0)) -> InstanceRaw:
    """instance.rs:16-25 Instance::to_raw."""
    ir = InstanceRaw()
    load_library().wr_instance_to_raw(C.byref(ir), _fptr(position, 3), _fptr(rotation_wxyz, 4), _fptr(scale, 3))
    return ir


def global_uniforms(time: float, use_shadowmaps: int) -> GlobalUniforms:
    return GlobalUniforms(float(time), 0, int(use_shadowmaps), 0)


def gather_layout(width: int, height: int, world: int, rank: int):
    """(bins owned by `rank`, bins per slot of the root's receive area): wr_gather_layout."""
    owned, slot = C.c_uint32(), C.c_uint32()
    load_library().wr_gather_layout(width, height, world, rank, C.byref(owned), C.byref(slot))
    return int(owned.value), int(slot.value)


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    rc = load_library().wr_comm_unique_id(buf)
    if rc != WR_OK:
        raise WrError(rc, "wr_comm_unique_id failed (is librccl.so.1 loadable?)")
    return bytes(buf)


def comm_create(uid: bytes, rank: int, world: int, device: int) -> int:
    buf = (C.c_uint8 * 128)(*uid)
    out = _P()
    rc = load_library().wr_comm_create(buf, rank, world, device, C.byref(out))
    if rc != WR_OK:
        raise WrError(rc, "wr_comm_create failed")
    return out.value


def comm_destroy(comm: int) -> None:
    load_library().wr_comm_destroy(comm)


def comm_transport(comm: int) -> "Transport":
    """wr_comm_transport: the grouped RCCL send / recv wr_gather_bins uses, as a WrTransport."""
    tp = Transport()
    rc = load_library().wr_comm_transport(comm, C.byref(tp))
    if rc != WR_OK:
        raise WrError(rc, "wr_comm_transport failed")
    return tp


def comm_library_path() -> str:
    """File name of the RCCL library the loader bound (wr_comm_library_path)."""
    buf = C.create_string_buffer(1024)
    rc = load_library().wr_comm_library_path(buf, len(buf))
    if rc != WR_OK:
        raise WrError(rc, "wr_comm_library_path failed (is librccl.so.1 loadable?)")
    return buf.value.decode()


class Glb:
    """A GLB file parsed by the library's C++ importer (wr_glb_*): the meshes, textures and
    materials load_model_gltf (resources.rs:17-136) would create, as numpy views."""

    def __init__(self, data: Optional[bytes] = None, path: Optional[str] = None):
        self.lib = load_library()
        h = _P()
        err = C.create_string_buffer(512)
        if path is not None:
            rc = self.lib.wr_glb_load_file(path.encode(), C.byref(h), err, len(err))
        else:
            rc = self.lib.wr_glb_parse(data, len(data), C.byref(h), err, len(err))
        if rc != WR_OK:
            raise WrError(rc, err.value.decode())
        self.h = h
        n = [C.c_uint32() for _ in range(3)]
        self.lib.wr_glb_counts(h, *[C.byref(x) for x in n])
        self.mesh_count, self.material_count, self.texture_count = (int(x.value) for x in n)

    def mesh(self, i: int):
        v, idx = _P(), _P()
        nv, ni, mat = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = self.lib.wr_glb_mesh(self.h, i, C.byref(v), C.byref(nv), C.byref(idx), C.byref(ni), C.byref(mat))
        if rc != WR_OK:
            raise WrError(rc, "bad mesh index")
        va = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), (nv.value, 14)).copy() if nv.value else np.zeros((0, 14), np.float32)
        ia = np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_uint32)), (ni.value,)).copy() if ni.value else np.zeros(0, np.uint32)
        return va, ia, int(mat.value)

    def texture(self, i: int):
        px = _P()
        w, h, srgb = C.c_uint32(), C.c_uint32(), C.c_int()
        rc = self.lib.wr_glb_texture(self.h, i, C.byref(px), C.byref(w), C.byref(h), C.byref(srgb))
        if rc != WR_OK:
            raise WrError(rc, "bad texture index")
        a = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), (h.value, w.value, 4)).copy()
        return a, bool(srgb.value)

    def material(self, i: int):
        b, n, m = C.c_uint32(), C.c_uint32(), C.c_uint32()
        met, rough = C.c_float(), C.c_float()
        rc = self.lib.wr_glb_material(self.h, i, C.byref(b), C.byref(n), C.byref(m), C.byref(met), C.byref(rough))
        if rc != WR_OK:
            raise WrError(rc, "bad material index")
        return int(b.value), int(n.value), int(m.value), float(met.value), float(rough.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.wr_glb_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hardware_queues_short_of(frames_in_flight: int):
    """HIP gives a process GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says
    otherwise, read at the first HIP call).  A context keeps three streams per frame in flight and
    one more; with fewer queues than that, frames in flight beyond one share queues and run slower
    than one frame at a time.  Returns the queue count if it is too small for `frames_in_flight`,
    else None."""
    if frames_in_flight <= 1:
        return None
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "") or 4)
    except ValueError:
        queues = 4
    return queues if queues < 3 * frames_in_flight + 1 else None


class Renderer:
    """One wr context (one GPU, one stream): the object a host keeps where the reference
    keeps its wgpu Device/Queue/pipelines (state.rs:30-66)."""

    def __init__(self, width: int, height: int, shadow_map_size: int = 2048, device: int = 0,
                 tile_rank: int = 0, tile_world: int = 1, stream: Optional[int] = None, frames_in_flight: int = 1,
                 color_format: int = 0):
        self.lib = load_library()
        queues = hardware_queues_short_of(frames_in_flight)
        if queues is not None:
            import warnings
            warnings.warn(f"frames_in_flight={frames_in_flight} with GPU_MAX_HW_QUEUES={queues}: streams will share hardware queues and "
                          f"frames in flight will run slower than one at a time; export GPU_MAX_HW_QUEUES=16 before the first HIP call",
                          RuntimeWarning, stacklevel=2)
        cfg = Config(width, height, shadow_map_size, color_format, device, tile_rank, tile_world, stream, frames_in_flight, 0)
        h = _P()
        rc = self.lib.wr_create(C.byref(cfg), C.byref(h))
        if rc != WR_OK:
            raise WrError(rc, "wr_create failed (is a HIP device visible?)")
        self.h = h
        self.width, self.height, self.shadow_map_size = width, height, shadow_map_size or 2048

    def _check(self, rc: int):
        if rc != WR_OK:
            raise WrError(rc, (self.lib.wr_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.wr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def upload_texture(self, rgba: np.ndarray, srgb: bool) -> int:
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        out = C.c_uint32()
        self._check(self.lib.wr_upload_texture(self.h, a.ctypes.data, a.shape[1], a.shape[0], int(srgb), C.byref(out)))
        return out.value

    def upload_material(self, model: int, base: int, normal: int, mr: int, metallic: float, roughness: float) -> int:
        out = C.c_uint32()
        self._check(self.lib.wr_upload_material(self.h, model, base, normal, mr, metallic, roughness, C.byref(out)))
        return out.value

    def upload_mesh(self, model: int, vertices: np.ndarray, indices: np.ndarray, material: int):
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        i = np.ascontiguousarray(indices, dtype=np.uint32).ravel()
        assert v.ndim == 2 and v.shape[1] == VERTEX_FLOATS
        self._check(self.lib.wr_upload_mesh(self.h, model, v.ctypes.data, v.shape[0], i.ctypes.data, i.size, material))

    def set_instance(self, model: int, inst: InstanceRaw):
        self._check(self.lib.wr_set_instances(self.h, model, C.byref(inst), 1))

    def upload_scene(self, scene) -> None:
        """Uploads a scene.Scene: textures, then per model materials, meshes and instance,
        in the order resources.rs:31-131 produces them."""
        self._check(self.lib.wr_clear_scene(self.h))
        for rgba, srgb in scene.textures:
            self.upload_texture(rgba, srgb)
        for model_id, model, inst in ((MODEL_GEOMETRY, scene.geometry, scene.geometry_instance),
                                      (MODEL_FOG_VOLUME, scene.fog, scene.fog_instance),
                                      (MODEL_LIGHT_MARKER, scene.marker, None)):
            if model is None:
                continue
            for m in model.materials:
                self.upload_material(model_id, m.base, m.normal, m.mr, m.metallic, m.roughness)
            for mesh in model.meshes:
                self.upload_mesh(model_id, mesh.vertices, mesh.indices, mesh.material)
            if inst is not None:
                self.set_instance(model_id, inst)

    # ---- frames
    def render_frame(self, camera: CameraUniform, light: LightUniform, globals_: GlobalUniforms, passes: int):
        self._check(self.lib.wr_render_frame(self.h, C.byref(camera), C.byref(light), C.byref(globals_), passes))

    def render_frame_static(self, camera, light, globals_, passes: int, device_color: int):
        """wr_render_frame_static: the same frame without any device allocation, written to the
        caller's device buffer (width*height*4 bytes)."""
        self._check(self.lib.wr_render_frame_static(self.h, C.byref(camera), C.byref(light), C.byref(globals_), passes, device_color))

    def upload_glb(self, model: int, glb: "Glb"):
        """load_model_gltf's uploads for one model: textures, materials, meshes in draw order."""
        self._check(self.lib.wr_upload_glb(self.h, model, glb.h))

    def render_batch(self, cameras: Sequence[CameraUniform], light, globals_, passes, shadows_every_frame=False,
                     device_frames: Optional[int] = None):
        arr = (CameraUniform * len(cameras))(*cameras)
        self._check(self.lib.wr_render_batch(self.h, arr, len(cameras), C.byref(light), C.byref(globals_), passes,
                                             int(shadows_every_frame), device_frames))

    def sync(self):
        self._check(self.lib.wr_sync(self.h))

    def resize(self, w: int, h: int):
        self._check(self.lib.wr_resize(self.h, w, h))
        if w and h:
            self.width, self.height = w, h

    # ---- results
    def read_color(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self.lib.wr_read_color(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_depth(self) -> np.ndarray:
        out = np.empty((self.height, self.width), np.float32)
        self._check(self.lib.wr_read_depth(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_visibility(self) -> np.ndarray:
        out = np.empty((self.height, self.width), np.uint32)
        self._check(self.lib.wr_read_visibility(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_shadow_map(self) -> np.ndarray:
        s = self.shadow_map_size
        out = np.empty((6, s, s), np.float32)
        self._check(self.lib.wr_read_shadow_map(self.h, out.ctypes.data, out.nbytes))
        return out

    def color_device_ptr(self) -> int:
        return int(self.lib.wr_color_device_ptr(self.h) or 0)

    def pack_owned_bins(self, device_dst: int) -> int:
        n = C.c_uint32()
        self._check(self.lib.wr_pack_owned_bins(self.h, device_dst, C.byref(n)))
        return n.value

    def scatter_bins(self, device_src: int, rank: int, world: int):
        self._check(self.lib.wr_scatter_bins(self.h, device_src, rank, world))

    def gather_bins_via(self, transport: "Transport", root: int = 0):
        self._check(self.lib.wr_gather_bins_via(self.h, C.byref(transport), root))

    def gather_bins(self, comm: int, root: int = 0):
        """wr_gather_bins: grouped RCCL sends of this rank's packed bins to `root`, queued behind
        the frame on the context's stream."""
        self._check(self.lib.wr_gather_bins(self.h, comm, root))

    def set_profiling(self, on: bool):
        self._check(self.lib.wr_set_profiling(self.h, int(on)))

    def set_counting(self, on: bool):
        self._check(self.lib.wr_set_counting(self.h, int(on)))

    def stats(self) -> FrameStats:
        st = FrameStats()
        self._check(self.lib.wr_get_stats(self.h, C.byref(st)))
        return st
