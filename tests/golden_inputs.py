"""Inputs shared by scripts/make_golden.py and the tests that check against tests/golden/."""
import os

import numpy as np

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def unit_inputs():
    rng = np.random.default_rng(0x5EED)
    sin_x = np.concatenate([
        np.array([0.0, -0.0, 1e-30, 0.5, 1.0, 3.14159, -3.14159, 100.0, 1e6, -4.5e6, 1.0e9, 2.0e9, np.inf], np.float32),
        rng.uniform(-10, 10, 40).astype(np.float32), rng.uniform(-5e6, 5e6, 60).astype(np.float32)])
    points = rng.uniform(-3000, 3000, (48, 3)).astype(np.float32)
    brdf = []
    for _ in range(24):
        n, l, v = (rng.normal(size=3) for _ in range(3))
        n, l, v = n / np.linalg.norm(n), l / np.linalg.norm(l), v / np.linalg.norm(v)
        hv = (l + v) / np.linalg.norm(l + v)
        brdf.append(np.concatenate([n, l, v, hv, rng.uniform(0, 1, 3), [rng.uniform(0.02, 1)], [rng.uniform(0, 1)]]).astype(np.float32))
    ambient = [np.concatenate([rng.uniform(0, 1, 3), [250000.0], [rng.uniform(1, 2000)], [rng.uniform(-1, 1)]]).astype(np.float32) for _ in range(16)]
    depth = np.concatenate([[0.0, 0.5, 0.9, 0.999, 0.9999, 1.0], rng.uniform(0.9, 1.0, 10)]).astype(np.float32)
    linear = np.concatenate([[-1.0, 0.0, 1e-6, 0.0031308, 0.5, 0.999, 1.0, 2.0, np.nan], rng.uniform(0, 1, 64)]).astype(np.float32)
    tris = []
    f, n = 3000.0, 1.0
    fl = 1 / np.tan(np.radians(55) / 2)
    for i in range(48):
        P = rng.uniform(-1, 1, (3, 3)) * 3 + np.array([0, 0, rng.uniform(-4, 2)])
        clip = []
        for k in range(3):
            x, y, z = P[k] * rng.choice([1, 1, 1, 50])
            clip += [fl / 1.777 * x, fl * y, (f + n) / (n - f) * z + 2 * f * n / (n - f), -z]
        tris.append((np.array(clip, np.float32).tolist(), int(i % 2)))
    return dict(sin=sin_x, points=points, brdf=brdf, ambient=ambient, depth=depth, linear=linear, triangles=tris)


def cube_textures():
    """Decoded images of the reference's Cube.glb: image 0 from the committed PNG, image 1 is
    the constant texel recorded in cube_glb.json."""
    import json
    from PIL import Image
    meta = json.load(open(os.path.join(GOLDEN, "cube_glb.json")))
    img0 = np.asarray(Image.open(os.path.join(GOLDEN, "cube_basecolor.png")).convert("RGBA"), np.uint8).copy()
    img1 = np.empty(tuple(meta["image1_shape"]), np.uint8)
    img1[...] = np.array(meta["image1_unique_texels"][0], np.uint8)
    return [img0, img1]


def cube_config1(w=1280, h=720):
    """BASELINE.json configs[0]: Cube.glb as geometry, one point light, PBS only
    (use_shadowmaps = 0, no fog), light marker on; instance scale 100, camera pitch -15."""
    sc = scene.cube_scene(cube_textures())
    cam = abi.camera_uniform((-500.0, 150.0, 0.0), -15.0, 0.0, 55.0, w, h)
    return sc, cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 0), abi.PASS_LIGHT_DEBUG, w, h, 256


def standin_small(w=192, h=108):
    sc = scene.sponza_standin(detail=0.08, tex_size=32)
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], w, h)
    return sc, cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 1), abi.PASS_ALL, w, h, 256
