"""A file of Sponza's shape and size through the native importer (wr_glb_* / wr_upload_glb,
standing for load_model_gltf, src/core/resources.rs:17-136, and Mesh::from_gltf,
src/core/mesh.rs:13-112).  The real res/models/Sponza.glb is not in the reference checkout;
tests/sponza_like.py writes a file with its structure from the procedural stand-in.

JPEG texels are the one place where the two importers may differ (the native decoder against
PIL's libjpeg, a few codes; neither is pinned against the reference's jpeg-decoder 0.3.0), so
frames are compared bit for bit on the native decoder's texels and by PSNR on PIL's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, gltf_loader, scene
from sponza_like import sponza_like_glb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both_importers(path, info):
    model, textures = gltf_loader.load_glb(path)
    g = abi.Glb(path=path)
    assert (g.mesh_count, g.material_count, g.texture_count) == (len(model.meshes), len(model.materials), len(textures))
    assert g.mesh_count == info["primitives"] and g.material_count == info["materials"] == 25
    for i, m in enumerate(model.meshes):  # one mesh per primitive, in document order (mesh.rs:20-21,102-108)
        v, idx, mat = g.mesh(i)
        assert np.array_equal(v, m.vertices) and np.array_equal(idx, m.indices) and mat == m.material, i
    assert [g.material(i) for i in range(g.material_count)] == [(m.base, m.normal, m.mr, m.metallic, m.roughness) for m in model.materials]
    native, exact, worst = [], 0, 0
    for i, (px, srgb) in enumerate(textures):
        a, s = g.texture(i)
        assert a.shape == px.shape and s == srgb, i
        d = int(np.abs(a.astype(np.int16) - px.astype(np.int16)).max())
        exact += d == 0
        worst = max(worst, d)
        native.append((a, s))
    g.close()
    return model, textures, native, exact, worst


def test_sponza_shaped_file_through_both_importers(tmp_path):
    """CPU: structure of the generated file and agreement of the two importers on it (reduced
    triangle count; the GPU test takes the full size)."""
    glb, info = sponza_like_glb(detail=0.15, tex_size=64)
    assert 90 <= info["primitives"] <= 110 and info["u16"] > 20 and info["u32"] > 20 and info["negative_handedness"] > 20
    assert info["images"] < 3 * info["materials"]  # images are shared between materials
    assert {"png_rgba", "png_rgb", "png_palette", "jpeg", "jpeg_444", "jpeg_progressive"} <= set(info["encodings"])
    p = tmp_path / "sponza_like.glb"
    p.write_bytes(glb)
    model, textures, native, exact, worst = _both_importers(str(p), info)
    assert sum(m.indices.size // 3 for m in model.meshes) == info["triangles"]
    assert exact >= len(textures) // 2   # every PNG texel is the same in both
    assert worst <= 8                    # JPEG: the two decoders stay within a few codes


@pytest.mark.gpu
def test_sponza_shaped_file_at_full_size(tmp_path):
    """260k triangles in a hundred primitives at 1920x1080 with every pass: the frame of
    wr_upload_glb equals the frame of the Python loader's scene (on the same texels) and the
    oracle's, bit for bit; with PIL's JPEG texels it stays above the PSNR bar; bench.py
    --sponza reports the file's counts and a frame rate."""
    from oracle_lib import Oracle
    from parity import compare
    glb, info = sponza_like_glb(detail=1.0, tex_size=256)
    assert 250_000 <= info["triangles"] <= 280_000 and 90 <= info["primitives"] <= 110
    p = tmp_path / "sponza_like.glb"
    p.write_bytes(glb)
    model, textures, native, exact, worst = _both_importers(str(p), info)
    w, h, s = 1920, 1080, 2048
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], w, h)
    light, glob = abi.light_at_time(2.0), abi.global_uniforms(2.0, 1)
    geom_inst = abi.instance_raw(scene.REFERENCE_GEOMETRY_INSTANCE["position"])
    fog_inst = abi.instance_raw(scene.REFERENCE_FOG_INSTANCE["position"], (1, 0, 0, 0), scene.REFERENCE_FOG_INSTANCE["scale"])
    cube = scene.cube_model().meshes[0]

    # the native importer's uploads (textures, materials, meshes in draw order), as a host without Python makes them
    r = abi.Renderer(w, h, s)
    g = abi.Glb(path=str(p))
    r.upload_glb(abi.MODEL_GEOMETRY, g)
    g.close()
    for model_id in (abi.MODEL_FOG_VOLUME, abi.MODEL_LIGHT_MARKER):
        r.upload_mesh(model_id, cube.vertices, cube.indices, 0)
    r.set_instance(abi.MODEL_GEOMETRY, geom_inst)
    r.set_instance(abi.MODEL_FOG_VOLUME, fog_inst)
    r.render_frame(cam, light, glob, abi.PASS_ALL)
    got = dict(color=r.read_color(), vis=r.read_visibility(), depth=r.read_depth(), shadow=r.read_shadow_map())
    r.close()
    assert (got["vis"] != 0xFFFFFFFF).mean() > 0.9

    def frame_of(tex, renderer):
        sc = scene.Scene(tex, model, scene.cube_model(), scene.cube_model(), geom_inst, fog_inst)
        renderer.upload_scene(sc)
        renderer.render_frame(cam, light, glob, abi.PASS_ALL, **({"threads": 16} if isinstance(renderer, Oracle) else {}))
        out = dict(color=renderer.read_color(), vis=renderer.read_visibility(), depth=renderer.read_depth(), shadow=renderer.read_shadow_map())
        renderer.close()
        return out

    py = frame_of(native, abi.Renderer(w, h, s))
    orc = frame_of(native, Oracle(w, h, s))
    for other in (py, orc):
        for k in ("vis", "depth", "shadow", "color"):
            assert np.array_equal(got[k], other[k]), k
    pil = frame_of(textures, abi.Renderer(w, h, s))
    m = compare(got["vis"], pil["vis"], got["color"], pil["color"])
    # (JPEG texels differ by up to `worst` codes between the two decoders.  That keeps the frame far
    # inside the PSNR bar, but a normal-map texel a few codes off moves a sharp highlight, so single
    # pixels change by a lot: no bound on the largest error is asserted for PIL's texels.)
    assert m["id_match"] == 1.0 and m["psnr"] >= 45.0 and m["exact_pixels"] >= 0.9, m

    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sponza", str(p), "--steps", "10", "--warmup", "2",
                        "--no-cpu-baseline", "--skip-march-alone"], capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    line = json.loads(b.stdout.strip().splitlines()[-1])
    assert f"P={info['primitives']} M=25" in b.stderr and f"I={3 * info['triangles']}" in b.stderr
    assert line["config"]["primitives"] == info["primitives"] and line["config"]["triangles"] == info["triangles"]
    assert line["config"]["scene"] == "sponza_like.glb" and line["value"] > 100.0 and line["scale_workloads"] is None
