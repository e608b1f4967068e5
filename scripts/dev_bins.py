"""Developer helper: per-bin load of the bench frame (needs a library built with -DWR_DEV)."""
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
r.render_frame(cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 1), abi.PASS_ALL)
r.sync()
lib = r.lib
lib.wr_dev_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
lib.wr_dev_read.restype = ctypes.c_int
def read(which, dtype=np.uint32):
    n = lib.wr_dev_read(r.h, which, None, 0)
    a = np.zeros(n, np.uint32)
    lib.wr_dev_read(r.h, which, a.ctypes.data, a.nbytes)
    return a
for name, g, T, layers in (("shadow", 0, S, 6), ("geometry", 1, (W, H), 1)):
    cnt = read(0 + g); bb = read(2 + g).reshape(-1, 2); off = read(4 + g); items = read(6 + g)
    tw, th = (T, T) if isinstance(T, int) else T
    bx, by = (tw + 63) // 64, (th + 63) // 64
    nb = bx * by
    stride = len(bb) // layers
    area = np.zeros(len(cnt), np.float64)
    big = np.zeros(len(cnt), np.int64)
    for gb in np.nonzero(cnt)[0]:
        layer, b = divmod(int(gb), nb)
        ri = items[off[gb]:off[gb] + cnt[gb]].astype(np.int64) + layer * stride
        mn, mx = bb[ri, 0], bb[ri, 1]
        ox, oy = (b % bx) * 64, (b // bx) * 64
        x0 = np.maximum(mn & 0xffff, ox); x1 = np.minimum(mx & 0xffff, ox + 63)
        y0 = np.maximum(mn >> 16, oy); y1 = np.minimum(mx >> 16, oy + 63)
        a = (x1.astype(np.int64) - x0 + 1).clip(0) * (y1.astype(np.int64) - y0 + 1).clip(0)
        area[gb] = a.sum(); big[gb] = (a > 128).sum()
    order = np.argsort(-area)
    print(f"== {name}: bins {len(cnt)} nonempty {(cnt>0).sum()} items {cnt.sum()} area {area.sum():.3e} big items {big.sum()}")
    print("  count percentiles 50/90/99/max:", np.percentile(cnt, [50, 90, 99, 100]))
    print("  area  percentiles 50/90/99/max:", np.percentile(area, [50, 90, 99, 100]))
    print("  top bins by area (area, n, big):", [(int(area[i]), int(cnt[i]), int(big[i])) for i in order[:12]])
    o2 = np.argsort(-cnt.astype(np.int64))
    print("  top bins by count (area, n, big):", [(int(area[i]), int(cnt[i]), int(big[i])) for i in o2[:12]])
# per-bin raster durations (wall clock ticks, 100 MHz)
dbg = read(8).view(np.uint64).reshape(3, 40000)
nbg = ((W + 63) // 64) * ((H + 63) // 64)
cntg = read(1)
for mode, name, nb_ in ((0, "colour", nbg), (2, "layers", nbg), (1, "shadow", 6144)):
    t = dbg[mode, :nb_].astype(np.float64) / 100.0  # microseconds
    o = np.argsort(-t)
    print(f"== raster {name}: sum {t.sum():.0f} us, max {t.max():.1f}, p99 {np.percentile(t,99):.1f}, p90 {np.percentile(t,90):.1f}, median {np.median(t):.1f}")
    if mode != 1:
        print("   top (us, n items):", [(round(float(t[i]), 1), int(cntg[i])) for i in o[:16]])
    else:
        c = read(0)
        print("   top (us, n items):", [(round(float(t[i]), 1), int(c[i])) for i in o[:16]])
