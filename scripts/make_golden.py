"""Generates the committed fixtures under tests/golden/ from the CPU oracle:
  oracle_vectors.json      known-answer vectors of the shader arithmetic (sine, noise, BRDF,
                           ambient term, depth_to_linear, sRGB) and of triangle set-up
  cube_frame_160x90.npz    the single-cube scene (Cube.glb as geometry, PBS only) at reduced size
  standin_frame_192x108.npz  small Sponza stand-in frame with shadows + fog
The reference has no golden data of its own, so these pin the oracle against regressions
and give the GPU tests fixed expected outputs; they are outputs of this repository's
oracle, not of the reference.  Run from the repository root."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wgpu_renderer_amd  # noqa: E402,F401
from wgpu_renderer_amd import abi, scene  # noqa: E402
from oracle_lib import Oracle, f3, lib  # noqa: E402
from golden_inputs import cube_config1, standin_small, unit_inputs  # noqa: E402

L = lib()
G = os.path.join(ROOT, "tests", "golden")


def bits(x):
    return int(np.float32(x).view(np.uint32))


def main():
    ui = unit_inputs()
    vec = {"sin": [], "rand": [], "value_noise": [], "fbm": [], "fog_noise": [], "brdf": [], "ambient": [],
           "depth_to_linear": [], "srgb_encode": [], "srgb_decode": [], "setup": []}
    for x in ui["sin"]:
        vec["sin"].append([bits(x), bits(L.wro_sin(float(x)))])
    for p in ui["points"]:
        vec["rand"].append([[bits(v) for v in np.floor(p)], bits(L.wro_rand(f3(np.floor(p))))])
        vec["value_noise"].append([[bits(v) for v in p], bits(L.wro_value_noise(f3(p)))])
        vec["fbm"].append([[bits(v) for v in p], bits(L.wro_fbm(f3(p * 0.01)))])
        vec["fog_noise"].append([[bits(v) for v in p], bits(L.wro_fog_noise(f3(p), 2.0))])
    for b in ui["brdf"]:
        out = (C.c_float * 3)()
        L.wro_brdf(f3(b[0:3]), f3(b[3:6]), f3(b[6:9]), f3(b[9:12]), f3(b[12:15]), float(b[15]), float(b[16]), out)
        vec["brdf"].append([[bits(v) for v in b], [bits(v) for v in out]])
    for a in ui["ambient"]:
        out = (C.c_float * 3)()
        L.wro_ambient((C.c_float * 4)(*a[0:4]), float(a[4]), float(a[5]), out)
        vec["ambient"].append([[bits(v) for v in a], [bits(v) for v in out]])
    for d in ui["depth"]:
        vec["depth_to_linear"].append([bits(d), bits(L.wro_depth_to_linear(float(d), 1.0, 3000.0))])
    for x in ui["linear"]:
        vec["srgb_encode"].append([bits(x), int(L.wro_srgb_encode(float(x)))])
    vec["srgb_decode"] = [bits(L.wro_srgb_decode(c)) for c in range(256)]
    for clip, shadow in ui["triangles"]:
        out = (C.c_uint32 * (28 * 8))()
        n = L.wro_setup_triangle((C.c_float * 12)(*clip), 640, 360, shadow, out, 8)
        vec["setup"].append([[bits(v) for v in clip], shadow, list(out[: 28 * n])])
    json.dump(vec, open(os.path.join(G, "oracle_vectors.json"), "w"))

    for name, (sc, cam, light, glob, passes, w, h, s) in (("cube_frame_160x90", cube_config1(160, 90)),
                                                          ("standin_frame_192x108", standin_small(192, 108))):
        o = Oracle(w, h, s)
        o.upload_scene(sc)
        o.render_frame(cam, light, glob, passes, threads=4)
        np.savez_compressed(os.path.join(G, name + ".npz"), color=o.read_color(), depth=o.read_depth(),
                            visibility=o.read_visibility(),
                            shadow_face3=o.read_shadow_map()[3] if passes & abi.PASS_SHADOWS else np.zeros((1, 1), np.float32))
        print(name, o.counters())
        o.close()


if __name__ == "__main__":
    main()
