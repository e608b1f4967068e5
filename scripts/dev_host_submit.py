"""Developer helper: host time spent inside wr_render_frame (submission only) against the
frame time of the pipelined bench frame, with and without the fog pass."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene


def run(nfl, passes, label, sc):
    W, H, S = 1920, 1080, 2048
    r = abi.Renderer(W, H, S, frames_in_flight=nfl)
    r.upload_scene(sc)
    light = abi.light_at_time(2.0)
    glob = abi.global_uniforms(2.0, 1)
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], W, H)
    for _ in range(10):
        r.render_frame(cam, light, glob, passes)
    r.sync()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_frame(cam, light, glob, passes)
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print("%-28s in flight %d: submit %.3f ms/frame, frame %.3f ms (%.0f frames/s)" %
          (label, nfl, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, n / (t2 - t0)), flush=True)
    r.close()


def main():
    sc = scene.sponza_standin(detail=1.0, tex_size=1024)
    full = abi.PASS_SHADOWS | abi.PASS_FOG | abi.PASS_LIGHT_DEBUG
    if len(sys.argv) > 1:  # one case: nofog | geom | full [frames in flight]
        passes = {"nofog": abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, "geom": abi.PASS_LIGHT_DEBUG, "full": full}[sys.argv[1]]
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 4, passes, sys.argv[1], sc)
        return
    for nfl in (1, 2, 3, 4):
        run(nfl, full, "shadows + geometry + fog", sc)
        run(nfl, abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, "shadows + geometry", sc)
        run(nfl, abi.PASS_LIGHT_DEBUG, "geometry only", sc)


main()
