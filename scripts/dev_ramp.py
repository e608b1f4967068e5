"""Developer helper: host time of each of the first wr_render_frame calls after the device has
gone idle (four frames in flight), and the time 20 frames take from first call to completion:
the shape of the driver's `--steps 20` bracket."""
import os, sys, time
# usage: dev_ramp.py [frames in flight] [profiling level] [hardware queues]   (bench.py asks for 16; the box exports 4)
os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[3] if len(sys.argv) > 3 else "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene

W, H, S = 1920, 1080, 2048
sc = scene.sponza_standin(detail=1.0, tex_size=1024)
r_nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
r = abi.Renderer(W, H, S, frames_in_flight=r_nfr)
r.upload_scene(sc)
r.set_profiling(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
light, glob = abi.light_at_time(2.0), abi.global_uniforms(2.0, 1)
rc = scene.REFERENCE_CAMERA
cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], W, H)
for _ in range(10):
    r.render_frame(cam, light, glob, abi.PASS_ALL)
r.sync()
for rep in range(3):
    ts = [time.perf_counter()]
    for _ in range(20):
        r.render_frame(cam, light, glob, abi.PASS_ALL)
        ts.append(time.perf_counter())
    r.sync()
    end = time.perf_counter()
    calls = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
    print("calls ms:", " ".join(f"{c:.2f}" for c in calls))
    print(f"submitted after {1e3 * (ts[-1] - ts[0]):.2f} ms, complete after {1e3 * (end - ts[0]):.2f} ms = {20 / (end - ts[0]):.0f} frames/s", flush=True)
t0 = time.perf_counter()
for _ in range(300):
    r.render_frame(cam, light, glob, abi.PASS_ALL)
r.sync()
print(f"300 frames: {300 / (time.perf_counter() - t0):.0f} frames/s  (hardware queues {os.environ['GPU_MAX_HW_QUEUES']}, frames in flight {r_nfr})", flush=True)
r.close()
