"""Developer helper: how many fog_noise octave evaluations of the bench frame each noise level
takes, from its hash table or directly, in the view-march and light-march kernels (needs a
library built with -DWR_DEV: bash scripts/dev_variants.sh dev -DWR_DEV)."""
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
r.set_counting(True)
r.render_frame(cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 1), abi.PASS_ALL)
st = r.stats()
lib = r.lib
lib.wr_dev_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
lib.wr_dev_read.restype = ctypes.c_int
a = np.zeros(2 * 16 * 2, np.uint64)
lib.wr_dev_read(r.h, 10, a.ctypes.data, a.nbytes)
a = a.reshape(2, 16, 2)
print("items", st.fog_fragments, "evals read by the marches", st.fog_noise_evals, "evaluated view", st.fog_noise_evals_view, "light", st.fog_noise_evals_light)
for kind, name, n in ((0, "view", st.fog_noise_evals_view), (1, "light", st.fog_noise_evals_light)):
    print(f"== {name}: per level, share of the kernel's evaluations that run the octave (table / direct)")
    for l in range(16):
        print(f"  fbm{l // 8 + 1} octave {l % 8}: table {a[kind, l, 0] / max(n, 1):6.3f}  direct {a[kind, l, 1] / max(n, 1):6.3f}")
b = np.zeros(2 * 16 * 6, np.uint64)
lib.wr_dev_read(r.h, 12, b.ctypes.data, b.nbytes)
b = b.reshape(2, 16, 6)
for kind, name, n in ((0, "view", st.fog_noise_evals_view), (1, "light", st.fog_noise_evals_light)):
    print(f"== {name}: directly evaluated octaves per evaluation, by the lattice points their wavefront's direct lanes span (<=64 <=128 <=256 <=512 <=1024 more)")
    for l in range(16):
        if b[kind, l].sum():
            print(f"  fbm{l // 8 + 1} octave {l % 8}: " + "  ".join(f"{x / max(n, 1):6.3f}" for x in b[kind, l]))
    tot = b[kind].sum(axis=0) / max(n, 1)
    print("  all levels:     " + "  ".join(f"{x:6.3f}" for x in tot))
