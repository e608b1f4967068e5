"""scripts/render_frames.py, the headless stand-in for the reference's window loop
(window.rs:55-69: update, render, present), run for three frames of fly-control input."""
import ctypes as C
import math
import runpy
import struct
import sys
import zlib
from pathlib import Path

import numpy as np
import pytest

from wgpu_renderer_amd import abi, scene

ROOT = Path(__file__).resolve().parents[1]


def read_png(path):
    """Decoder for what the script writes: 8-bit RGBA, no interlace, filter 0 on every row."""
    data = Path(path).read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, size = 8, b"", None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
        if kind == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, interlace) == (8, 6, 0)
            size = (w, h)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    w, h = size
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 4)


def fly_update(pose, inp, speed, sensitivity, dt):
    """camera.rs:78-101 with get_vecs of camera.rs:61-76, in double precision."""
    x, y, z, pitch, yaw = pose
    dx, dy, fwd, rgt, up_ = inp
    pitch = min(max(pitch - dy * sensitivity * 0.022, -89.0), 89.0)
    yaw = math.fmod(yaw + dx * sensitivity * 0.022, 360.0)
    if yaw < 0.0:
        yaw += 360.0
    ys, yc = math.sin(math.radians(yaw)), math.cos(math.radians(yaw))
    ps, pc = math.sin(math.radians(pitch)), math.cos(math.radians(pitch))
    forward = np.array([pc * yc, ps, pc * ys])
    forward /= np.linalg.norm(forward)
    right = np.array([-ys, 0.0, yc])
    right /= np.linalg.norm(right)
    up = np.cross(right, forward)
    p = np.array([x, y, z]) + forward * fwd * speed * dt + right * rgt * speed * dt + up * up_ * speed * dt
    return [p[0], p[1], p[2], pitch, yaw]


@pytest.mark.gpu
def test_three_fly_control_frames_written_by_the_script_are_the_frames_the_library_renders(tmp_path, capsys):
    w, h, s, detail, tex = 320, 180, 256, 0.25, 64
    argv = ["--fly", "3", "--out", str(tmp_path), "--width", str(w), "--height", str(h), "--detail", str(detail),
            "--tex-size", str(tex), "--shadow-map-size", str(s)]
    module = runpy.run_path(str(ROOT / "scripts" / "render_frames.py"), run_name="render_frames")
    module["main"](argv)
    assert len(capsys.readouterr().out.strip().splitlines()) == 3
    files = sorted(tmp_path.glob("frame_*.png"))
    assert [f.name for f in files] == ["frame_0000.png", "frame_0001.png", "frame_0002.png"]

    # the same three steps of the controls, by the library's function and by the rules in double precision
    rc = scene.REFERENCE_CAMERA
    lib = abi.load_library()
    pose = (C.c_float * 5)(*rc["position"], rc["pitch"], rc["yaw"])
    inp = (C.c_float * 5)(3.0, 0.5, 1.0, 0.0, 0.0)
    exact = [*rc["position"], rc["pitch"], rc["yaw"]]
    r = abi.Renderer(w, h, s)
    r.upload_scene(scene.sponza_standin(detail=detail, tex_size=tex))
    frames = []
    for k in range(3):
        lib.wr_camera_fly_update(pose, inp, 400.0, 1.0, 1.0 / 60.0)
        exact = fly_update(exact, list(inp), 400.0, 1.0, 1.0 / 60.0)
        assert np.allclose(list(pose), exact, rtol=1e-6, atol=1e-4)
        t = 2.0 + k / 60.0
        cam = abi.camera_uniform((pose[0], pose[1], pose[2]), pose[3], pose[4], rc["fovy"], w, h)
        r.render_frame(cam, abi.light_at_time(t), abi.global_uniforms(t, 1), abi.PASS_ALL)
        frames.append(r.read_color())
    r.close()
    for k, f in enumerate(files):
        assert np.array_equal(read_png(f), frames[k]), f.name
    assert not np.array_equal(frames[0], frames[2])  # the camera moved
