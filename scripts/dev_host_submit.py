"""Developer helper: host time spent inside wr_render_frame (submission only) against the
frame time of the pipelined bench frame."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wgpu_renderer_amd
from wgpu_renderer_amd import abi, scene

def main():
    nfl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    W, H, S = 1920, 1080, 2048
    sc = scene.sponza_standin(detail=1.0, tex_size=256)
    r = abi.Renderer(W, H, S, frames_in_flight=nfl)
    r.upload_scene(sc)
    light = abi.light_at_time(2.0)
    glob = abi.global_uniforms(2.0, 1)
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], W, H)
    passes = abi.PASS_SHADOWS | abi.PASS_FOG | abi.PASS_LIGHT_DEBUG
    for _ in range(10):
        r.render_frame(cam, light, glob, passes)
    r.sync()
    n = 300
    t0 = time.perf_counter()
    sub = 0.0
    for _ in range(n):
        a = time.perf_counter()
        r.render_frame(cam, light, glob, passes)
        sub += time.perf_counter() - a
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print("frames in flight %d: submit %.3f ms/frame inside the call, loop %.3f ms/frame, with final sync %.3f ms/frame" % (nfl, sub / n * 1e3, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
main()
