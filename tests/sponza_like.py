"""A GLB with the shape of the Khronos Sponza the reference loads (src/core/state.rs:380
`models/Sponza.glb`, which is not part of the reference checkout): about a hundred primitives
over 25 materials, images shared between materials, PNG (RGBA, RGB, palette) next to baseline
and progressive JPEG, 16-bit and 32-bit indices, negative tangent handedness, and the things
load_model_gltf ignores present in the file (node transforms, alphaMode, doubleSided,
emissive factors, names, a sampler).  Geometry and images are the procedural stand-in's
(wgpu_renderer_amd.scene.sponza_standin), cut into primitives.  Test-side only."""
import io
import json
import struct

import numpy as np

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import scene


def _encode(img, kind):
    from PIL import Image
    buf = io.BytesIO()
    if kind == "png_rgba":
        Image.fromarray(img, "RGBA").save(buf, format="PNG")
    elif kind == "png_rgb":
        Image.fromarray(img[..., :3].copy(), "RGB").save(buf, format="PNG")
    elif kind == "png_palette":
        Image.fromarray(img[..., :3].copy(), "RGB").quantize(64).save(buf, format="PNG")
    elif kind == "jpeg":
        Image.fromarray(img[..., :3].copy(), "RGB").save(buf, format="JPEG", quality=92, subsampling=2)
    elif kind == "jpeg_444":
        Image.fromarray(img[..., :3].copy(), "RGB").save(buf, format="JPEG", quality=95, subsampling=0)
    elif kind == "jpeg_progressive":
        Image.fromarray(img[..., :3].copy(), "RGB").save(buf, format="JPEG", quality=90, progressive=True)
    else:
        raise ValueError(kind)
    return buf.getvalue(), ("image/jpeg" if kind.startswith("jpeg") else "image/png")


def sponza_like_glb(detail=1.0, tex_size=256, primitives=100):
    """Returns (glb bytes, info).  info: primitive / material / image counts and how many
    primitives use each index width and image encoding."""
    sc = scene.sponza_standin(detail=detail, tex_size=tex_size)
    geo = sc.geometry
    tris = np.array([m.indices.size // 3 for m in geo.meshes], np.float64)
    # primitives per stand-in mesh: in proportion to its triangles, at least one each
    parts = np.maximum(1, np.floor(tris / tris.sum() * (primitives - len(tris)) + 1).astype(int))
    blobs, views, accessors = [], [], []

    def add_view(raw, target=None):
        off = sum(len(b) for b in blobs)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if target:
            v["target"] = target
        blobs.append(raw + b"\0" * (-len(raw) % 4))
        views.append(v)
        return len(views) - 1

    def add_accessor(arr, ctype, atype, target, minmax=False):
        a = {"bufferView": add_view(np.ascontiguousarray(arr).tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]), "type": atype}
        if minmax:
            a["min"], a["max"] = [float(x) for x in arr.min(axis=0)], [float(x) for x in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    info = {"u16": 0, "u32": 0, "negative_handedness": 0}
    gl_meshes, pi = [], 0
    for mi, m in enumerate(geo.meshes):
        tri = m.indices.reshape(-1, 3)
        prims = []
        for chunk in np.array_split(np.arange(len(tri)), parts[mi]):
            if len(chunk) == 0:
                continue
            used, local = np.unique(tri[chunk].ravel(), return_inverse=True)
            v = m.vertices[used]
            flip = pi % 3 == 1  # TANGENT = (-t, -1): tangent * w is t again (mesh.rs:56-58)
            tan = np.concatenate([-v[:, 8:11] if flip else v[:, 8:11], np.full((len(v), 1), -1.0 if flip else 1.0, np.float32)], axis=1)
            info["negative_handedness"] += int(flip)
            small = len(used) <= 65535 and pi % 2 == 1
            info["u16" if small else "u32"] += 1
            idx = local.astype(np.uint16 if small else np.uint32)
            prims.append({"attributes": {"POSITION": add_accessor(v[:, 0:3], 5126, "VEC3", 34962, True),
                                         "NORMAL": add_accessor(v[:, 5:8], 5126, "VEC3", 34962),
                                         "TANGENT": add_accessor(tan.astype(np.float32), 5126, "VEC4", 34962),
                                         "TEXCOORD_0": add_accessor(v[:, 3:5], 5126, "VEC2", 34962)},
                          "indices": add_accessor(idx, 5123 if small else 5125, "SCALAR", 34963), "material": int(m.material), "mode": 4})
            pi += 1
        gl_meshes.append({"name": f"part_{mi}", "primitives": prims})

    # images: every material brings three; the later materials borrow the normal and the
    # metallic-roughness image of an earlier one (shared images, as Sponza's materials do)
    images, image_of_texture, encodings = [], {}, {}
    opaque_kinds = ["jpeg", "png_rgb", "jpeg_progressive", "jpeg_444", "png_palette"]

    def image_for(tex_id, role, k):
        if tex_id in image_of_texture:
            return image_of_texture[tex_id]
        rgba = sc.textures[tex_id][0]
        if role == "base" and (rgba[..., 3] != 255).any():
            kind = "png_rgba"
        elif role == "mr":
            kind = "png_rgb"  # (roughness and metalness are read exactly: no lossy encoding)
        else:
            kind = opaque_kinds[k % len(opaque_kinds)]
        raw, mime = _encode(rgba, kind)
        images.append({"bufferView": add_view(raw), "mimeType": mime, "name": f"{role}_{tex_id}_{kind}"})
        encodings[kind] = encodings.get(kind, 0) + 1
        image_of_texture[tex_id] = len(images) - 1
        return image_of_texture[tex_id]

    gl_textures, gl_materials = [], []

    def texture_for(image):
        gl_textures.append({"source": image, "sampler": 0})
        return {"index": len(gl_textures) - 1}

    for k, mat in enumerate(geo.materials):
        donor = geo.materials[k % 5] if k >= 13 else mat
        base = image_for(mat.base, "base", k)
        normal = image_for(donor.normal, "normal", k + 1)
        mr = image_for(donor.mr, "mr", k)
        entry = {"name": f"material {k}", "doubleSided": k % 4 == 0, "emissiveFactor": [0.0, 0.1 * (k % 3), 0.0],
                 "pbrMetallicRoughness": {"baseColorTexture": texture_for(base), "metallicRoughnessTexture": texture_for(mr),
                                          "metallicFactor": float(mat.metallic), "roughnessFactor": float(mat.roughness),
                                          "baseColorFactor": [1.0, 1.0, 1.0, 1.0]},
                 "normalTexture": texture_for(normal)}
        if (sc.textures[mat.base][0][..., 3] != 255).any():
            entry["alphaMode"] = "MASK" if k % 2 else "BLEND"
        gl_materials.append(entry)

    nodes = [{"mesh": i, "name": f"node_{i}", "translation": [3.0 * i, -7.0, 11.0], "scale": [0.008, 0.008, 0.008]} for i in range(len(gl_meshes))]
    js = {"asset": {"version": "2.0", "generator": "tests/sponza_like.py"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}],
          "nodes": nodes, "meshes": gl_meshes, "materials": gl_materials, "textures": gl_textures, "images": images,
          "samplers": [{"magFilter": 9729, "minFilter": 9987, "wrapS": 10497, "wrapT": 10497}],
          "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": sum(len(b) for b in blobs)}]}
    jb = json.dumps(js).encode()
    jb += b" " * (-len(jb) % 4)
    bb = b"".join(blobs)
    glb = (struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(jb) + 8 + len(bb)) + struct.pack("<I4s", len(jb), b"JSON") + jb
           + struct.pack("<I4s", len(bb), b"BIN\0") + bb)
    info.update(primitives=pi, materials=len(gl_materials), images=len(images), encodings=encodings,
                triangles=int(tris.sum()), vertices=int(sum(accessors[p["attributes"]["POSITION"]]["count"] for m in gl_meshes for p in m["primitives"])))
    return glb, info
