"""GLB / glTF 2.0 import that reproduces the reference's loader, quirks included
(src/core/resources.rs:17-136 load_model_gltf, src/core/mesh.rs:13-112 Mesh::from_gltf):

  * meshes are taken from `document.meshes()`, node transforms are ignored
    (resources.rs:31-33); one Mesh per primitive, in document order (mesh.rs:20-21);
  * tangent = TANGENT.xyz * TANGENT.w, bitangent = normal x tangent (mesh.rs:56-64);
    a primitive without TANGENT gets NaN tangents/bitangents, because calc_tangents runs
    before the indices are read and then divides by a zero count (mesh.rs:69,151-155);
  * a material's missing texture slot falls back to image 0 (resources.rs:46,74,101);
  * RGB images are padded to RGBA with alpha 255 (resources.rs:184-191); base colour is
    sRGB, normal and metallic-roughness are linear (resources.rs:62,89,116);
  * primitive.material defaults to 0 (mesh.rs:23).

Image decoding uses PIL where the reference uses the `image` crate; both sides of a parity
test are fed the texels this loader produced, so decoder differences stay outside the
compared path.
"""
from __future__ import annotations

import io
import json
import struct
from typing import List, Tuple

import numpy as np

from .scene import Material, Mesh, Model

_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _parse_glb(data: bytes):
    magic, version, _length = struct.unpack_from("<4sII", data, 0)
    if magic != b"glTF" or version != 2:
        raise ValueError("not a glTF 2.0 binary")
    off, js, bin_chunk = 12, None, b""
    while off < len(data):
        clen, ctype = struct.unpack_from("<I4s", data, off)
        chunk = data[off + 8: off + 8 + clen]
        if ctype == b"JSON":
            js = json.loads(chunk)
        elif ctype == b"BIN\x00":
            bin_chunk = chunk
        off += 8 + clen
    if js is None:
        raise ValueError("GLB without a JSON chunk")
    return js, bin_chunk


def _accessor(js, buffers, index) -> np.ndarray:
    a = js["accessors"][index]
    dt, size = _COMP[a["componentType"]]
    n = _NCOMP[a["type"]]
    bv = js["bufferViews"][a["bufferView"]]
    buf = buffers[bv["buffer"]]
    start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or size * n
    count = a["count"]
    if stride == size * n:
        arr = np.frombuffer(buf, dtype=dt, count=count * n, offset=start).reshape(count, n)
    else:
        raw = np.frombuffer(buf, dtype=np.uint8, count=stride * (count - 1) + size * n, offset=start)
        rows = np.lib.stride_tricks.as_strided(raw, shape=(count, size * n), strides=(stride, 1))
        arr = np.ascontiguousarray(rows).view(dt).reshape(count, n)
    if a.get("normalized", False) and a["componentType"] != 5126:
        info = np.iinfo(np.dtype(dt))
        arr = np.maximum(arr.astype(np.float32) / float(info.max), -1.0)
    return arr


def _decode_image(raw: bytes) -> np.ndarray:
    """Returns (h, w, 4) uint8 RGBA.  RGB (and palette images, which the `image` crate
    expands to RGB) get alpha 255; other channel counts are rejected like
    resources.rs:51."""
    from PIL import Image

    im = Image.open(io.BytesIO(raw))
    if im.mode == "P":
        im = im.convert("RGBA" if "transparency" in im.info else "RGB")
    if im.mode == "RGB":
        rgb = np.asarray(im, dtype=np.uint8)
        return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
    if im.mode == "RGBA":
        return np.asarray(im, dtype=np.uint8).copy()
    raise ValueError(f"unsupported image channel layout {im.mode!r} (reference asserts 3 or 4 channels)")


def load_glb(path: str) -> Tuple[Model, List[Tuple[np.ndarray, bool]]]:
    """Returns (model, textures); texture ids inside the model's materials index into the
    returned list, whose entries are (rgba8, srgb)."""
    data = open(path, "rb").read()
    js, bin_chunk = _parse_glb(data)
    buffers = [bin_chunk]  # GLB-embedded buffer 0; external buffers are not needed by the reference's assets
    meshes: List[Mesh] = []
    for mesh in js.get("meshes", []):
        for prim in mesh["primitives"]:
            attrs = prim["attributes"]
            pos = _accessor(js, buffers, attrs["POSITION"]).astype(np.float32)
            nrm = _accessor(js, buffers, attrs["NORMAL"]).astype(np.float32)
            n = pos.shape[0]
            v = np.zeros((n, 14), np.float32)
            v[:, 0:3] = pos
            v[:, 5:8] = nrm
            if "TANGENT" in attrs:
                t4 = _accessor(js, buffers, attrs["TANGENT"]).astype(np.float32)
                tan = t4[:, :3] * t4[:, 3:4]
                v[:, 8:11] = tan
                v[:, 11:14] = np.cross(nrm, tan).astype(np.float32)
            else:
                v[:, 8:14] = np.nan
            v[:, 3:5] = _accessor(js, buffers, attrs["TEXCOORD_0"]).astype(np.float32)
            idx = _accessor(js, buffers, prim["indices"]).astype(np.uint32).ravel()
            meshes.append(Mesh(v, idx, int(prim.get("material", 0))))

    images: List[np.ndarray] = []
    for img in js.get("images", []):
        if "bufferView" in img:
            bv = js["bufferViews"][img["bufferView"]]
            o = bv.get("byteOffset", 0)
            images.append(_decode_image(bin_chunk[o: o + bv["byteLength"]]))
        else:
            raise ValueError("external image URIs are not supported")

    textures: List[Tuple[np.ndarray, bool]] = []
    cache = {}

    def tex_id(image_index: int, srgb: bool) -> int:
        key = (image_index, srgb)
        if key not in cache:
            cache[key] = len(textures)
            textures.append((images[image_index], srgb))
        return cache[key]

    def source(texinfo) -> int:
        if texinfo is None:
            return 0
        return int(js["textures"][texinfo["index"]].get("source", 0))

    materials: List[Material] = []
    for m in js.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        materials.append(Material(
            base=tex_id(source(pbr.get("baseColorTexture")), True),
            normal=tex_id(source(m.get("normalTexture")), False),
            mr=tex_id(source(pbr.get("metallicRoughnessTexture")), False),
            metallic=float(pbr.get("metallicFactor", 1.0)),
            roughness=float(pbr.get("roughnessFactor", 1.0)),
        ))
    return Model(meshes, materials), textures
