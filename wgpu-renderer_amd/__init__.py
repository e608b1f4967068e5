"""MI355X-native implementation of the per-frame render path of nullprop/wgpu-renderer.

The hot path (state.rs:630-809 + res/shaders/*.wgsl) lives in csrc/ as hand-written gfx950
HIP kernels behind the C ABI of include/wr_abi.h; this package is the Python host side:
`abi` binds the library with ctypes, `scene` builds the benchmark inputs, `gltf_loader`
imports GLB files the way the reference does, `sharding` splits frames across ranks.
"""
from . import abi, scene, sharding  # noqa: F401
from .abi import Renderer, WrError  # noqa: F401
