"""Developer helper: where the bins of the three raster walks spend their time on the bench frame
(needs a library built with -DWR_DEV: bash scripts/dev_variants.sh dev -DWR_DEV).  Ticks are
wall_clock64 (100 MHz) of thread 0 of each bin's workgroup, summed over the bins."""
import ctypes, os, sys
import numpy as np
os.environ.setdefault("WR_HIP_LIBRARY", os.path.abspath("build_dev/libwr_dev.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_renderer_amd as wr
from wgpu_renderer_amd import abi, scene
W, H, S = 1920, 1080, 2048
r = wr.Renderer(W, H, shadow_map_size=S, device=0)
r.upload_scene(scene.sponza_standin(tex_size=256))
cam = abi.camera_uniform((-500, 150, 0), 0.0, 0.0, 55.0, W, H)
for _ in range(3):
    r.render_frame(cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 1), abi.PASS_ALL)
    r.sync()
lib = r.lib
lib.wr_dev_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
lib.wr_dev_read.restype = ctypes.c_int
a = np.zeros(3 * 8, np.uint64)
lib.wr_dev_read(r.h, 14, a.ctypes.data, a.nbytes)
a = a.reshape(3, 8).astype(np.float64)
for mode, name in enumerate(("colour", "shadow", "layers")):
    bins = max(a[mode, 4], 1)
    us = a[mode, :4] / bins / 100.0
    print(f"{name:7s}: bins/frame {bins / 3:.0f}  chunks/bin {a[mode, 5] / bins:.2f}  large triangles/bin {a[mode, 6] / bins:.1f}  "
          f"mean us per bin: init {us[0]:.2f}  loads+lane path {us[1]:.2f}  wavefront path {us[2]:.2f}  write-out {us[3]:.2f}  total {us.sum():.2f}")
b = np.zeros(8, np.uint64)
lib.wr_dev_read(r.h, 16, b.ctypes.data, b.nbytes)
b = b.astype(np.float64) / 3
print(f"shadow walk per frame: wavefront path visits {b[0]:.0f} 8x8 blocks, {b[1]:.0f} ({100 * b[1] / max(b[0], 1):.1f} %) change no texel (as drawn: what a per-block "
      f"farthest-depth test could skip at most), {b[2]:.0f} ({100 * b[2] / max(b[0], 1):.1f} %) are fully covered; lane path draws {b[3]:.0f} triangles, "
      f"{b[4]:.0f} ({100 * b[4] / max(b[3], 1):.1f} %) change no texel")
