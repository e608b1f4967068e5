"""Extracts the data of the reference's res/models/Cube.glb that the build needs without
the reference present:
  wgpu-renderer_amd/assets/cube_mesh.npz   36 ModelVertex rows + indices (fog volume, marker)
  tests/golden/cube_basecolor.png          image 0 of the GLB, re-encoded losslessly
  tests/golden/cube_glb.json               image sizes, image 1's constant texel, material factors
Run from the repository root in a container that has /root/reference."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wgpu_renderer_amd  # noqa: E402
from wgpu_renderer_amd import gltf_loader  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/res/models/Cube.glb"
model, textures = gltf_loader.load_glb(src)
assert len(model.meshes) == 1
m = model.meshes[0]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
np.savez(os.path.join(root, "wgpu-renderer_amd", "assets", "cube_mesh.npz"), vertices=m.vertices, indices=m.indices)
from PIL import Image  # noqa: E402

img0 = textures[model.materials[0].base][0]
img1 = textures[model.materials[0].mr][0]
Image.fromarray(img0, "RGBA").save(os.path.join(root, "tests", "golden", "cube_basecolor.png"), optimize=True)
uniq = np.unique(img1.reshape(-1, 4), axis=0)
meta = {
    "image0_shape": list(img0.shape), "image1_shape": list(img1.shape),
    "image1_unique_texels": uniq.tolist(),
    "metallic_factor": model.materials[0].metallic, "roughness_factor": model.materials[0].roughness,
    "normal_slot_falls_back_to_image0": model.materials[0].normal != model.materials[0].base,
    "vertices": int(m.vertices.shape[0]), "indices": int(m.indices.size),
}
json.dump(meta, open(os.path.join(root, "tests", "golden", "cube_glb.json"), "w"), indent=1)
print(meta)
