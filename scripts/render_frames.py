#!/usr/bin/env python3
"""Headless frame writer: renders the flythrough (or the reference's default pose) through
libwr_hip.so and writes the frames as PNG files, where the reference presents them to a
window surface (window.rs:63-69, state.rs:806).

    python scripts/render_frames.py --out frames --poses 8 [--sponza PATH] [--width 1920 --height 1080]

The camera path is scene.flythrough_pose (circle of radius 400 at y = 150, looking along the
tangent); --fly N instead integrates N steps of the reference's fly controls
(wr_camera_fly_update, camera.rs:78-101) from the default pose with a constant forward input
and a slow mouse drift.
"""
import argparse
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_renderer_amd  # noqa: E402,F401
from wgpu_renderer_amd import abi, scene  # noqa: E402


def write_png(path, rgba):
    h, w = rgba.shape[:2]
    raw = b"".join(b"\0" + rgba[y].tobytes() for y in range(h))

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--poses", type=int, default=8)
    ap.add_argument("--fly", type=int, default=0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--time", type=float, default=2.0)
    ap.add_argument("--shadow-map-size", type=int, default=2048)
    ap.add_argument("--sponza", default="")
    args = ap.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    W, H = args.width, args.height
    r = abi.Renderer(W, H, args.shadow_map_size)
    if args.sponza:
        from wgpu_renderer_amd import gltf_loader
        model, textures = gltf_loader.load_glb(args.sponza)
        sc = scene.Scene(textures, model, scene.cube_model(), scene.cube_model(),
                         abi.instance_raw(scene.REFERENCE_GEOMETRY_INSTANCE["position"]),
                         abi.instance_raw(scene.REFERENCE_FOG_INSTANCE["position"], (1, 0, 0, 0), scene.REFERENCE_FOG_INSTANCE["scale"]))
    else:
        sc = scene.sponza_standin(detail=args.detail, tex_size=args.tex_size)
    r.upload_scene(sc)
    rc = scene.REFERENCE_CAMERA
    if args.fly:
        import ctypes as C
        pose = (C.c_float * 5)(*rc["position"], rc["pitch"], rc["yaw"])
        inp = (C.c_float * 5)(3.0, 0.5, 1.0, 0.0, 0.0)
        poses = []
        for _ in range(args.fly):
            r.lib.wr_camera_fly_update(pose, inp, 400.0, 1.0, 1.0 / 60.0)
            poses.append(((pose[0], pose[1], pose[2]), pose[3], pose[4]))
    else:
        poses = [scene.flythrough_pose(k, args.poses) for k in range(args.poses)]
    for k, (pos, pitch, yaw) in enumerate(poses):
        t = args.time + k / 60.0
        cam = abi.camera_uniform(pos, pitch, yaw, rc["fovy"], W, H)
        r.render_frame(cam, abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL)
        write_png(os.path.join(args.out, f"frame_{k:04d}.png"), r.read_color())
        print(f"frame {k}: pos=({pos[0]:.1f},{pos[1]:.1f},{pos[2]:.1f}) pitch={pitch:.1f} yaw={yaw:.1f}", flush=True)
    r.close()


if __name__ == "__main__":
    main()
