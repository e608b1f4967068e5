"""CPU-only checks of the boundary: the library loads, exports every symbol include/wr_abi.h
declares, struct layouts match the reference's #[repr(C)] structs, and the host scene math
(camera.rs / light.rs / instance.rs mirrors) agrees with closed-form values."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "wr_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = abi.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in wr_abi.h but not exported"
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)


def test_struct_layouts_match_reference():
    # model.rs:14-22, instance.rs:9-14, camera.rs:104-112, light.rs:11-18, state.rs:21-28
    assert abi.VERTEX_FLOATS * 4 == 56
    assert C.sizeof(abi.InstanceRaw) == 100
    assert C.sizeof(abi.CameraUniform) == 224
    assert C.sizeof(abi.LightUniform) == 416
    assert abi.LightUniform.color.offset == 16 and abi.LightUniform.matrices.offset == 32
    assert C.sizeof(abi.GlobalUniforms) == 16
    assert abi.CameraUniform.position.offset == 192 and abi.CameraUniform.planes.offset == 208


def test_create_without_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(abi.WrError):
        abi.Renderer(64, 64, 64)


def _m(a):
    return np.array(a, dtype=np.float64).reshape(4, 4).T  # column-major -> math layout


def test_camera_uniform_matches_closed_form():
    w, h = 1920, 1080
    cu = abi.camera_uniform((-500.0, 150.0, 0.0), 0.0, 0.0, 55.0, w, h)
    view, proj = _m(cu.view), _m(cu.proj)
    # yaw 0, pitch 0: forward +x, right +z, up +y (camera.rs:61-76)
    p = view @ np.array([-500.0 + 10.0, 150.0, 0.0, 1.0])
    assert np.allclose(p[:3], [0, 0, -10], atol=1e-4)  # 10 units ahead = -z in view space
    f = 1.0 / np.tan(np.radians(55.0) / 2)
    assert np.isclose(proj[0, 0], f / (w / h), rtol=1e-6) and np.isclose(proj[1, 1], f, rtol=1e-6)
    n, fa = 1.0, 3000.0
    assert np.isclose(proj[2, 2], (fa + n) / (n - fa), rtol=1e-6)
    assert np.isclose(proj[2, 3], 2 * fa * n / (n - fa), rtol=1e-6)
    assert proj[3, 2] == -1.0
    assert list(cu.planes) == [1.0, 3000.0, float(w), float(h)]
    assert list(cu.position) == [-500.0, 150.0, 0.0, 1.0]
    inv = _m(cu.inv_view_proj)
    ref = np.linalg.inv(proj @ view)
    assert np.allclose(inv, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_camera_pitch_yaw_vectors():
    cu = abi.camera_uniform((0.0, 0.0, 0.0), 30.0, 90.0, 55.0, 100, 100)
    view = _m(cu.view)
    fwd = np.array([np.cos(np.radians(30)) * np.cos(np.radians(90)), np.sin(np.radians(30)), np.cos(np.radians(30)) * np.sin(np.radians(90))])
    assert np.allclose(view[:3, :3] @ fwd, [0, 0, -1], atol=1e-6)


def test_light_matrices_and_animation():
    lu = abi.light_at_time(2.0)
    # state.rs:610-617
    assert np.isclose(lu.position[0], np.sin(1.0) * 500, rtol=1e-6)
    assert np.isclose(lu.position[1], 250 + np.sin(0.6) * 200, rtol=1e-6)
    assert np.isclose(lu.position[2], np.sin(1.6) * 100, rtol=1e-6)
    assert np.isclose(lu.color[0], abs(np.sin(2.0)), rtol=1e-6) and lu.color[3] == 250000.0
    pos = np.array(list(lu.position), dtype=np.float64)
    dirs = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]  # light.rs:35-40
    for i, d in enumerate(dirs):
        m = _m(lu.matrices[i])
        q = m @ np.append(pos + 100.0 * np.array(d), 1.0)
        ndc = q[:3] / q[3]
        assert abs(ndc[0]) < 1e-4 and abs(ndc[1]) < 1e-4 and q[3] > 0  # looks down its axis
        # 90 degree frustum: a point at 45 degrees lands on the clip edge
        up = {0: (0, 1, 0), 1: (0, 1, 0), 2: (1, 0, 0), 3: (-1, 0, 0), 4: (0, 1, 0), 5: (0, 1, 0)}[i]
        q = m @ np.append(pos + 100.0 * np.array(d) + 100.0 * np.array(up), 1.0)
        assert np.isclose(q[1] / q[3], 1.0, atol=1e-4)
    # t = 0 gives a black light (|sin 0| = 0)
    z = abi.light_at_time(0.0)
    assert list(z.color)[:3] == [0.0, 0.0, 0.0]


def test_instance_to_raw_is_T_S_R():
    ir = abi.instance_raw((0.0, 30.0, 0.0), (1, 0, 0, 0), (1360.0, 30.0, 600.0))  # state.rs:421-425
    m = _m(ir.model)
    assert np.allclose(m, [[1360, 0, 0, 0], [0, 30, 0, 30], [0, 0, 600, 0], [0, 0, 0, 1]])
    assert np.allclose(np.array(ir.normal).reshape(3, 3), np.eye(3))
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)  # 45 degrees about y
    ir = abi.instance_raw((1.0, 2.0, 3.0), (c, 0.0, s, 0.0), (2.0, 3.0, 4.0))
    R = np.array(ir.normal, dtype=np.float64).reshape(3, 3).T
    assert np.allclose(R @ [1, 0, 0], [np.cos(np.pi / 4), 0, -np.sin(np.pi / 4)], atol=1e-6)
    m = _m(ir.model)
    assert np.allclose(m[:3, :3], np.diag([2, 3, 4]) @ R, atol=1e-6) and np.allclose(m[:3, 3], [1, 2, 3])


def test_camera_fly_update_matches_reference_rules():
    lib = abi.load_library()
    pose = (C.c_float * 5)(0.0, 0.0, 0.0, 80.0, 350.0)
    inp = (C.c_float * 5)(300.0, -600.0, 1.0, 0.0, 0.0)
    lib.wr_camera_fly_update(pose, inp, 400.0, 2.0, 0.5)
    assert pose[3] == 89.0                      # pitch clamp, camera.rs:80-84
    assert np.isclose(pose[4], (350.0 + 300 * 2 * 0.022) % 360.0, atol=1e-4)  # yaw wrap, camera.rs:85-89
    assert np.isclose(np.linalg.norm([pose[0], pose[1], pose[2]]), 200.0, rtol=1e-5)  # speed*dt


def test_cube_asset_matches_reference_glb():
    ref = "/root/reference/res/models/Cube.glb"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    from wgpu_renderer_amd import gltf_loader
    model, textures = gltf_loader.load_glb(ref)
    cube = scene.cube_model().meshes[0]
    assert np.array_equal(model.meshes[0].vertices, cube.vertices)
    assert np.array_equal(model.meshes[0].indices, cube.indices)
    v = cube.vertices
    assert v.shape == (36, 14) and np.abs(v[:, 0:3]).max() <= 1.000002
    # bitangent = normal x tangent (mesh.rs:61-64)
    assert np.allclose(v[:, 11:14], np.cross(v[:, 5:8], v[:, 8:11]))
    mat = model.materials[0]
    assert textures[mat.base][1] is True and textures[mat.normal][1] is False
    assert textures[mat.normal][0] is textures[mat.base][0]  # missing normal map -> image 0 (resources.rs:74)
    assert np.unique(textures[mat.mr][0].reshape(-1, 4), axis=0).tolist() == [[0, 20, 0, 255]]


def test_gltf_loader_without_tangents_gives_nan(tmp_path):
    # mesh.rs:67-69,151-155: calc_tangents runs on an empty index list and divides by zero
    import json
    import struct
    from wgpu_renderer_amd import gltf_loader
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nrm = np.array([[0, 0, 1]] * 3, np.float32)
    uv = np.array([[0, 0], [1, 0], [0, 1]], np.float32)
    idx = np.array([0, 1, 2], np.uint16)
    blobs = [pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes() + b"\0\0"]
    views, off = [], 0
    for b in blobs:
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(b)})
        off += len(b)
    js = {
        "asset": {"version": "2.0"}, "buffers": [{"byteLength": off}], "bufferViews": views,
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC3"},
            {"bufferView": 2, "componentType": 5126, "count": 3, "type": "VEC2"},
            {"bufferView": 3, "componentType": 5123, "count": 3, "type": "SCALAR"}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3}]}],
    }
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    bb = b"".join(blobs)
    glb = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(jb) + 8 + len(bb)) + struct.pack("<I4s", len(jb), b"JSON") + jb + struct.pack("<I4s", len(bb), b"BIN\0") + bb
    p = tmp_path / "t.glb"
    p.write_bytes(glb)
    model, textures = gltf_loader.load_glb(str(p))
    v = model.meshes[0].vertices
    assert np.isnan(v[:, 8:14]).all() and model.meshes[0].material == 0
    assert model.meshes[0].indices.dtype == np.uint32 and model.meshes[0].indices.tolist() == [0, 1, 2]


def test_standin_scene_is_deterministic_and_sized():
    a = scene.sponza_standin(detail=0.1, tex_size=16)
    b = scene.sponza_standin(detail=0.1, tex_size=16)
    assert a.geometry.triangle_count == b.geometry.triangle_count > 1000
    for ma, mb in zip(a.geometry.meshes, b.geometry.meshes):
        assert np.array_equal(ma.vertices, mb.vertices) and np.array_equal(ma.indices, mb.indices)
    for (ta, sa), (tb, sb) in zip(a.textures, b.textures):
        assert sa == sb and np.array_equal(ta, tb)
    assert len(a.geometry.materials) == 25 and len(a.textures) == 75
    # some materials carry alpha < 1 texels (ordered-blend path), tangent frames are orthonormal
    assert any((t[..., 3] < 255).any() for t, s in a.textures if s)
    v = a.geometry.meshes[2].vertices
    assert np.allclose(np.linalg.norm(v[:, 5:8], axis=1), 1, atol=1e-4)
    assert np.allclose(np.sum(v[:, 5:8] * v[:, 8:11], axis=1), 0, atol=1e-4)


# ------------------------------------------------------------------ C++ GLB importer
def _png(rgba: np.ndarray, mode: str) -> bytes:
    """Minimal PNG writer for the importer tests: mode 'rgb', 'rgba' or 'palette' (with tRNS),
    every scanline with a different filter type so that all five unfilter paths run."""
    import struct
    import zlib
    h, w = rgba.shape[:2]
    chunks = []
    if mode == "palette":
        colours, inv = np.unique(rgba.reshape(-1, 4), axis=0, return_inverse=True)
        assert len(colours) <= 256
        rows = inv.reshape(h, w).astype(np.uint8)
        ihdr = struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)
        chunks += [(b"PLTE", colours[:, :3].astype(np.uint8).tobytes()), (b"tRNS", colours[:, 3].astype(np.uint8).tobytes())]
        bpp = 1
    else:
        ch = 3 if mode == "rgb" else 4
        rows = rgba[:, :, :ch].reshape(h, w * ch)
        ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 6, 0, 0, 0)
        bpp = ch
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        f = y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - a
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            enc = cur - pred
        raw.append(f)
        raw += (enc & 255).astype(np.uint8).tobytes()
        prev = cur
    chunks = [(b"IHDR", ihdr)] + chunks + [(b"IDAT", zlib.compress(bytes(raw))), (b"IEND", b"")]
    out = b"\x89PNG\r\n\x1a\n"
    for t, body in chunks:
        out += struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    return out


def _synthetic_glb(rng, with_tangent=True):
    import json
    import struct
    n = 12
    pos = rng.standard_normal((n, 3)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    uv16 = rng.integers(0, 65536, (n, 2)).astype(np.uint16)       # normalised u16 texture coordinates
    tan = np.concatenate([rng.standard_normal((n, 3)), rng.choice([-1.0, 1.0], (n, 1))], axis=1).astype(np.float32)
    idx8 = rng.integers(0, n, 18).astype(np.uint8)
    idx32 = rng.integers(0, n, 9).astype(np.uint32)
    interleaved = np.concatenate([pos, nrm], axis=1).astype(np.float32)  # one view, byteStride 24
    imgs = [rng.integers(0, 256, (5, 7, 4)).astype(np.uint8), rng.integers(0, 256, (4, 4, 4)).astype(np.uint8),
            rng.integers(0, 4, (6, 3, 4)).astype(np.uint8) * 80]
    pngs = [_png(imgs[0], "rgba"), _png(imgs[1], "rgb"), _png(imgs[2], "palette")]
    imgs[1] = imgs[1].copy()
    imgs[1][..., 3] = 255
    blobs = [interleaved.tobytes(), uv16.tobytes(), tan.tobytes(), idx8.tobytes(), idx32.tobytes()] + pngs
    views, off = [], 0
    for k, b in enumerate(blobs):
        b += b"\0" * (-len(b) % 4)
        blobs[k] = b
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(b)}
        if k == 0:
            v["byteStride"] = 24
        views.append(v)
        off += len(b)
    attrs = {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}
    if with_tangent:
        attrs["TANGENT"] = 3
    js = {
        "asset": {"version": "2.0"}, "buffers": [{"byteLength": off}], "bufferViews": views,
        "accessors": [
            {"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": n, "type": "VEC3"},
            {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": n, "type": "VEC3"},
            {"bufferView": 1, "componentType": 5123, "normalized": True, "count": n, "type": "VEC2"},
            {"bufferView": 2, "componentType": 5126, "count": n, "type": "VEC4"},
            {"bufferView": 3, "componentType": 5121, "count": 18, "type": "SCALAR"},
            {"bufferView": 4, "componentType": 5125, "count": 9, "type": "SCALAR"}],
        "images": [{"bufferView": 5, "mimeType": "image/png"}, {"bufferView": 6, "mimeType": "image/png"}, {"bufferView": 7, "mimeType": "image/png"}],
        "textures": [{"source": 1}, {"source": 2}, {"source": 0}],
        "materials": [
            {"name": "a \u00e9 \"quoted\"", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.25, "roughnessFactor": 0.5},
             "normalTexture": {"index": 1}},
            {"pbrMetallicRoughness": {"metallicRoughnessTexture": {"index": 2}}},
            {}],
        "meshes": [{"primitives": [{"attributes": attrs, "indices": 4, "material": 1}, {"attributes": attrs, "indices": 5}]},
                   {"primitives": [{"attributes": attrs, "indices": 5, "material": 2}]}],
        "nodes": [{"mesh": 0, "translation": [5, 6, 7]}],
    }
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    bb = b"".join(blobs)
    glb = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(jb) + 8 + len(bb)) + struct.pack("<I4s", len(jb), b"JSON") + jb + struct.pack("<I4s", len(bb), b"BIN\0") + bb
    return glb, imgs


@pytest.mark.parametrize("with_tangent", [True, False])
def test_native_glb_importer_matches_python_loader(tmp_path, with_tangent):
    """wr_glb_* (C++: JSON, accessors with strides / normalised integers, PNG with every filter
    type, palette + tRNS) against the Python loader (numpy + PIL) on a synthetic file."""
    from wgpu_renderer_amd import gltf_loader
    glb, imgs = _synthetic_glb(np.random.default_rng(0x61B), with_tangent)
    p = tmp_path / "s.glb"
    p.write_bytes(glb)
    model, textures = gltf_loader.load_glb(str(p))
    for g in (abi.Glb(data=glb), abi.Glb(path=str(p))):
        assert (g.mesh_count, g.material_count, g.texture_count) == (len(model.meshes), len(model.materials), len(textures))
        assert g.mesh_count == 3 and g.material_count == 3
        for i, m in enumerate(model.meshes):
            v, idx, mat = g.mesh(i)
            assert np.array_equal(v, m.vertices, equal_nan=True) and np.array_equal(idx, m.indices) and mat == m.material
            assert np.isnan(v[:, 8:14]).all() == (not with_tangent)
        for i, (px, srgb) in enumerate(textures):
            a, s = g.texture(i)
            assert np.array_equal(a, px) and s == srgb
        for i, m in enumerate(model.materials):
            b, n, mr, met, rough = g.material(i)
            assert (b, n, mr) == (m.base, m.normal, m.mr) and met == np.float32(m.metallic) and rough == np.float32(m.roughness)
        # decoded texels are the bytes that were encoded; image 0 is the fallback of empty slots
        assert np.array_equal(g.texture(g.material(0)[0])[0], imgs[1]) and np.array_equal(g.texture(g.material(0)[1])[0], imgs[2])
        assert np.array_equal(g.texture(g.material(2)[0])[0], imgs[0]) and g.texture(g.material(2)[0])[1] is True
        g.close()


def test_native_glb_importer_rejects_what_the_reference_rejects():
    import struct
    import zlib
    with pytest.raises(abi.WrError):
        abi.Glb(data=b"not a glb at all....")
    # a grey PNG has one channel: resources.rs:51 asserts 3 or 4
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    grey = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0ab\0cd")) + chunk(b"IEND", b"")
    import json
    js = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(grey)}], "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(grey)}],
          "images": [{"bufferView": 0}]}
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    bb = grey + b"\0" * (-len(grey) % 4)
    glb = struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(jb) + 8 + len(bb)) + struct.pack("<I4s", len(jb), b"JSON") + jb + struct.pack("<I4s", len(bb), b"BIN\0") + bb
    with pytest.raises(abi.WrError) as e:
        abi.Glb(data=glb)
    assert e.value.code == abi.WR_ERR_UNSUPPORTED and "3 or 4" in str(e.value)


def test_native_glb_importer_on_the_reference_cube():
    ref = "/root/reference/res/models/Cube.glb"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    g = abi.Glb(path=ref)
    cube = scene.cube_model().meshes[0]
    v, idx, mat = g.mesh(0)
    assert np.array_equal(v, cube.vertices) and np.array_equal(idx, cube.indices) and mat == 0
    from wgpu_renderer_amd import gltf_loader
    model, textures = gltf_loader.load_glb(ref)
    assert g.texture_count == len(textures)
    for i, (px, srgb) in enumerate(textures):
        a, s = g.texture(i)
        assert np.array_equal(a, px) and s == srgb


def _numeric_constants(text, decl):
    """name -> list of floats for every `<decl> NAME[...] = expr;` whose expr is numbers, '/', ',' only."""
    import re
    out = {}
    for m in re.finditer(decl + r"\s+(\w+)(?:\[\d+\])?\s*=\s*([^;]+);", text):
        expr = re.sub(r"vec2<f32>|[{}()]", " ", m.group(2))
        expr = re.sub(r"(?<=[0-9.])f\b", "", expr)
        if not re.fullmatch(r"[\d\s.,/+-]+", expr):
            continue
        vals = []
        for part in expr.split(","):
            if part.strip():
                num = [float(x) for x in part.split("/")]
                v = num[0]
                for d in num[1:]:
                    v /= d
                vals.append(v)
        out[m.group(1)] = vals
    return out


def test_wgsl_include_expansion(tmp_path):
    # preprocessor.rs:6-23: top-level includes only, the whole line with its newline replaced
    from wgpu_renderer_amd import wgsl
    (tmp_path / "a.wgsl").write_text("#include c.wgsl\n#include g.wgsl\nfn main() {}\n")
    (tmp_path / "c.wgsl").write_text("const K = 3;\n")
    (tmp_path / "g.wgsl").write_text("#include c.wgsl\nstruct G { x: f32 }\n")
    out = wgsl.preprocess_wgsl(str(tmp_path), "a.wgsl")
    # only the top file's include lines are expanded, in their order: c.wgsl is pasted first,
    # then g.wgsl, whose own "#include c.wgsl" line arrives after that replacement and stays
    assert out == "const K = 3;\n#include c.wgsl\nstruct G { x: f32 }\nfn main() {}\n"


def test_shader_constants_equal_the_reference_shaders():
    """Every constant of the reference's constants.wgsl, read from fog.wgsl and pbr.wgsl as the
    reference's preprocessor flattens them, is present under a recognisable name and with the
    same value in the device arithmetic and in the oracle."""
    shader_dir = "/root/reference/res/shaders"
    if not os.path.exists(os.path.join(shader_dir, "constants.wgsl")):
        pytest.skip("reference checkout not present")
    from wgpu_renderer_amd import wgsl as wgsl_pre
    fog_text = wgsl_pre.preprocess_wgsl(shader_dir, "fog.wgsl")
    pbr_text = wgsl_pre.preprocess_wgsl(shader_dir, "pbr.wgsl")
    assert "#include" not in fog_text and "#include" not in pbr_text
    wgsl = _numeric_constants(fog_text, r"const")
    assert wgsl == _numeric_constants(pbr_text, r"const")
    assert len(wgsl) >= 13 and wgsl["FOG_MAX_STEPS"] == [30.0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = {}
    for decl in (r"constexpr float", r"constexpr int"):
        dev.update(_numeric_constants(open(os.path.join(root, "wgpu-renderer_amd", "csrc", "wr_device.h")).read(), decl))
    # `constexpr float kA = 0.5f, kB = 0.1f;` declares two names in one statement
    dev_text = open(os.path.join(root, "wgpu-renderer_amd", "csrc", "wr_device.h")).read()
    import re
    m = re.search(r"kFogDensityColorX = ([\d.]+)f, kFogDensityColorY = ([\d.]+)f;", dev_text)
    dev["kFogDensityColor"] = [float(m.group(1)), float(m.group(2))]
    orc = {}
    for decl in (r"const float", r"const int"):
        orc.update(_numeric_constants(open(os.path.join(root, "oracle", "wr_oracle.cpp")).read(), decl))

    def camel(name):
        return "k" + "".join(p.capitalize() if not p.isdigit() else p for p in name.lower().split("_"))

    for name, val in wgsl.items():
        f32 = [float(np.float32(x)) for x in val]
        oname = "PI_C" if name == "PI" else name
        assert [float(np.float32(x)) for x in orc[oname]] == f32, name
        assert [float(np.float32(x)) for x in dev[camel(name)]] == f32, name


def _image_only_glb(image_bytes):
    """A GLB that holds nothing but one embedded image (and a material that names it)."""
    import json
    import struct
    bb = image_bytes + b"\0" * (-len(image_bytes) % 4)
    js = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(bb)}],
          "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(image_bytes)}],
          "images": [{"bufferView": 0}], "textures": [{"source": 0}],
          "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]}
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    return struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(jb) + 8 + len(bb)) + struct.pack("<I4s", len(jb), b"JSON") + jb + struct.pack("<I4s", len(bb), b"BIN\0") + bb


def _test_picture(rng, w, h):
    """Smooth gradients, a few hard edges and some noise: exercises DC, low and high AC."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 23.0), 127 + 120 * np.cos(y / 7.0), 40 + 200 * ((x // 16 + y // 16) % 2)], axis=2)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling,progressive,size", [
    ("4:4:4", False, (64, 48)), ("4:2:2", False, (67, 45)), ("4:2:0", False, (129, 70)),
    ("4:2:0", True, (96, 80)), ("4:4:4", True, (33, 17)), ("4:2:0", False, (1, 1)), ("4:2:2", True, (200, 9))])
def test_native_jpeg_decode_is_close_to_libjpeg(subsampling, progressive, size):
    """JPEG textures through the native importer against PIL (libjpeg).  The two decoders use
    different inverse transforms and upsampling filters, so texels agree closely, not bit for
    bit; the reference's own decoder (jpeg-decoder 0.3.0) is not in the checkout: unpinned."""
    import io
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    w, h = size
    src = _test_picture(rng, w, h)
    buf = io.BytesIO()
    PIL.fromarray(src, "RGB").save(buf, "JPEG", quality=92, subsampling=subsampling, progressive=progressive, optimize=progressive)
    data = buf.getvalue()
    want = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
    g = abi.Glb(data=_image_only_glb(data))
    got, srgb = g.texture(0)
    assert srgb and got.shape == (h, w, 4) and np.all(got[..., 3] == 255)
    diff = np.abs(got[..., :3].astype(np.int32) - want.astype(np.int32))
    assert diff.mean() < 0.5 and diff.max() <= 8, (diff.mean(), diff.max())
    # and both are decodes of the picture that was encoded
    assert np.abs(got[..., :3].astype(np.int32) - src.astype(np.int32)).mean() < np.abs(want.astype(np.int32) - src.astype(np.int32)).mean() + 0.5


def test_native_jpeg_decode_with_restart_markers_and_errors():
    import io
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    src = _test_picture(rng, 80, 56)
    buf = io.BytesIO()
    PIL.fromarray(src, "RGB").save(buf, "JPEG", quality=85, subsampling="4:2:0", restart_marker_blocks=3)
    data = buf.getvalue()
    assert b"\xff\xdd" in data  # DRI present
    want = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
    got, _ = abi.Glb(data=_image_only_glb(data)).texture(0)
    diff = np.abs(got[..., :3].astype(np.int32) - want.astype(np.int32))
    assert diff.mean() < 0.5 and diff.max() <= 8
    # a grey JPEG has one channel: resources.rs:51 asserts 3 or 4
    buf = io.BytesIO()
    PIL.fromarray(src[..., 0], "L").save(buf, "JPEG")
    with pytest.raises(abi.WrError) as e:
        abi.Glb(data=_image_only_glb(buf.getvalue()))
    assert e.value.code == abi.WR_ERR_UNSUPPORTED
    # truncated data is an error, not a crash
    with pytest.raises(abi.WrError):
        abi.Glb(data=_image_only_glb(data[: len(data) // 3]))


def test_native_jpeg_decode_survives_corrupt_streams():
    """Flipped bytes and cuts anywhere in a baseline and in a progressive stream: the importer
    either decodes something or reports an error; it never reads outside the buffer (the
    library runs inside the caller's process)."""
    import io
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(6)
    src = _test_picture(rng, 48, 40)
    for progressive in (False, True):
        buf = io.BytesIO()
        PIL.fromarray(src, "RGB").save(buf, "JPEG", quality=80, subsampling="4:2:0", progressive=progressive)
        good = bytearray(buf.getvalue())
        for trial in range(150):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            if trial % 3 == 0:
                bad = bad[: int(rng.integers(4, len(bad)))]
            try:
                g = abi.Glb(data=_image_only_glb(bytes(bad)))
                px, _ = g.texture(0)
                assert px.ndim == 3 and px.shape[2] == 4
            except abi.WrError as e:
                assert e.code in (abi.WR_ERR_INVALID_ARGUMENT, abi.WR_ERR_UNSUPPORTED)


def test_python_host_says_when_hardware_queues_are_too_few(monkeypatch):
    """Several frames in flight need a HIP hardware queue per stream (three per frame in flight
    and one more); HIP's default is four per process."""
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert abi.hardware_queues_short_of(1) is None
    assert abi.hardware_queues_short_of(2) == 4 and abi.hardware_queues_short_of(4) == 4
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert abi.hardware_queues_short_of(2) is None and abi.hardware_queues_short_of(4) == 8
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert abi.hardware_queues_short_of(4) is None
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "many")
    assert abi.hardware_queues_short_of(3) == 4
