"""Scene containers and the inputs of the benchmark / parity configurations.

  * `cube_model()`        -- the reference's res/models/Cube.glb mesh (36 vertices, 12
                             triangles) in ModelVertex layout, used as fog volume and light
                             marker exactly like state.rs:389-405.  The arrays ship as
                             assets/cube_mesh.npz (extracted by scripts/extract_cube.py).
  * `sponza_standin()`    -- procedural atrium with Sponza's rough extent, triangle count,
                             material count and texture sizes.  res/models/Sponza.glb is
                             not part of the reference checkout; pass a real file through
                             gltf_loader.load_glb when one is available.
  * `reference_*`         -- the constants State::new hard-codes (state.rs:116-122,134,
                             407-425).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass
class Mesh:
    vertices: np.ndarray  # (n, 14) float32: position3 uv2 normal3 tangent3 bitangent3
    indices: np.ndarray   # (3*t,) uint32
    material: int


@dataclass
class Material:
    base: int
    normal: int
    mr: int
    metallic: float = 1.0
    roughness: float = 1.0


@dataclass
class Model:
    meshes: List[Mesh] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)

    @property
    def triangle_count(self) -> int:
        return sum(m.indices.size // 3 for m in self.meshes)

    @property
    def vertex_count(self) -> int:
        return sum(m.vertices.shape[0] for m in self.meshes)


@dataclass
class Scene:
    textures: List[Tuple[np.ndarray, bool]]  # (rgba8 (h,w,4), srgb)
    geometry: Model
    fog: Optional[Model]
    marker: Optional[Model]
    geometry_instance: object = None  # abi.InstanceRaw
    fog_instance: object = None


# ------------------------------------------------------------------ reference constants
REFERENCE_CAMERA = dict(position=(-500.0, 150.0, 0.0), pitch=0.0, yaw=0.0, fovy=55.0)  # state.rs:116-122
REFERENCE_LIGHT_INTENSITY = 250000.0                                                    # state.rs:134
REFERENCE_GEOMETRY_INSTANCE = dict(position=(60.0, 0.0, 35.0), scale=(1.0, 1.0, 1.0))   # state.rs:407-412
REFERENCE_FOG_INSTANCE = dict(position=(0.0, 30.0, 0.0), scale=(1360.0, 30.0, 600.0))   # state.rs:421-425
SEED = 0x5EED


def cube_model() -> Model:
    """Cube.glb as the reference loads it (tangent*w, bitangent = n x t)."""
    d = np.load(os.path.join(_HERE, "assets", "cube_mesh.npz"))
    return Model([Mesh(d["vertices"].astype(np.float32), d["indices"].astype(np.uint32), 0)], [])


# ------------------------------------------------------------------ procedural geometry
def _finish(P: np.ndarray, UV: np.ndarray, nu: int, nv: int, flip: bool):
    """P: (nv+1, nu+1, 3) grid of positions, UV same grid -> (vertices, indices).  Normals
    come from the grid's central differences; tangent follows +u, bitangent = n x t."""
    dPu = np.gradient(P, axis=1)
    dPv = np.gradient(P, axis=0)
    n = np.cross(dPu, dPv)
    if flip:
        n = -n
    n /= np.maximum(np.linalg.norm(n, axis=2, keepdims=True), 1e-20)
    t = dPu - n * np.sum(dPu * n, axis=2, keepdims=True)
    t /= np.maximum(np.linalg.norm(t, axis=2, keepdims=True), 1e-20)
    b = np.cross(n, t)
    V = np.concatenate([P, UV, n, t, b], axis=2).reshape(-1, 14).astype(np.float32)
    j, i = np.mgrid[0:nv, 0:nu]
    a = (j * (nu + 1) + i).ravel()
    bq = a + 1
    c = a + (nu + 1)
    d = c + 1
    if not flip:
        tri = np.stack([a, bq, d, a, d, c], axis=1)
    else:
        tri = np.stack([a, d, bq, a, c, d], axis=1)
    return V, tri.reshape(-1).astype(np.uint32)


def _patch(origin, du, dv, nu, nv, uv_scale=(1.0, 1.0), flip=False):
    """Planar patch origin + s*du + t*dv; front face looks along du x dv."""
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t)
    P = np.asarray(origin, np.float64)[None, None, :] + S[..., None] * np.asarray(du, np.float64) + T[..., None] * np.asarray(dv, np.float64)
    UV = np.stack([S * uv_scale[0], T * uv_scale[1]], axis=2)
    return _finish(P, UV, nu, nv, flip)


def _surface(fn, nu, nv, uv_scale=(1.0, 1.0), flip=False):
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t)
    P = fn(S, T)
    UV = np.stack([S * uv_scale[0], T * uv_scale[1]], axis=2)
    return _finish(P, UV, nu, nv, flip)


def _merge(parts):
    vs, is_, base = [], [], 0
    for v, i in parts:
        vs.append(v)
        is_.append(i + base)
        base += v.shape[0]
    return np.concatenate(vs).astype(np.float32), np.concatenate(is_).astype(np.uint32)


def _box(lo, hi, seg=(1, 1, 1), uv=1.0):
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    sx, sy, sz = seg
    return _merge([
        _patch(lo + ez, ex, ey, sx, sy, (uv, uv)),                 # +z
        _patch(lo + ex, -ex, ey, sx, sy, (uv, uv)),                # -z
        _patch(lo + ex + ez, -ez, ey, sz, sy, (uv, uv)),           # +x
        _patch(lo, ez, ey, sz, sy, (uv, uv)),                      # -x
        _patch(lo + ey + ez, ex, -ez, sx, sz, (uv, uv)),           # +y
        _patch(lo, ex, ez, sx, sz, (uv, uv)),                      # -y
    ])


# ------------------------------------------------------------------ procedural textures
def _value_noise(rng, size, cells):
    g = rng.random((cells, cells))
    rep = max(size // cells, 1)
    a = np.repeat(np.repeat(g, rep, axis=0), rep, axis=1)[:size, :size]
    if a.shape[0] < size:
        a = np.pad(a, ((0, size - a.shape[0]), (0, size - a.shape[1])), mode="wrap")
    # cheap smoothing so that bilinear/nearest choices matter but the map is not blocky
    return (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(a, (1, 1), (0, 1))) * 0.25


def _material_textures(rng, size, kind, color, alpha_mode, rough, metal):
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    if kind == "brick":
        row = np.floor(y * 8)
        xx = x * 4 + 0.5 * (row % 2)
        h = np.minimum(np.abs(np.modf(xx)[0] - 0.5), np.abs(np.modf(y * 8)[0] - 0.5)) * 4.0
        h = np.clip(h, 0, 1)
    elif kind == "tile":
        h = np.clip(np.minimum(np.abs(np.modf(x * 6)[0] - 0.5), np.abs(np.modf(y * 6)[0] - 0.5)) * 6.0, 0, 1)
    elif kind == "fabric":
        h = 0.5 + 0.25 * np.sin(x * 2 * np.pi * 24) + 0.25 * np.sin(y * 2 * np.pi * 24)
    else:
        h = _value_noise(rng, size, max(size // 16, 2))
    h = 0.7 * h + 0.3 * _value_noise(rng, size, max(size // 8, 2))
    col = np.asarray(color, np.float64)[None, None, :] * (0.55 + 0.45 * h[..., None])
    col = np.clip(col + (rng.random((size, size, 1)) - 0.5) * 0.04, 0, 1)
    if alpha_mode == "cutout":
        a = ((np.sin(x * 2 * np.pi * 5) * np.sin(y * 2 * np.pi * 7) + 0.35 * np.sin((x + y) * 2 * np.pi * 3)) > 0.05).astype(np.float64)
    elif alpha_mode == "half":
        a = np.full((size, size), 128.0 / 255.0)
    else:
        a = np.ones((size, size))
    base = np.concatenate([col, a[..., None]], axis=2)
    base = np.round(base * 255).astype(np.uint8)
    gx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * size / 64.0
    gy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * size / 64.0
    n = np.stack([-gx, -gy, np.ones_like(h)], axis=2)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    normal = np.concatenate([n * 0.5 + 0.5, np.ones((size, size, 1))], axis=2)
    normal = np.round(normal * 255).astype(np.uint8)
    r = np.clip(rough + 0.15 * (h - 0.5), 0.04, 1.0)
    m = np.full_like(h, metal)
    mr = np.stack([np.zeros_like(h), r, m, np.ones_like(h)], axis=2)
    mr = np.round(mr * 255).astype(np.uint8)
    return base, normal, mr


_PALETTE = [
    # kind, colour, alpha mode, roughness, metalness
    ("tile", (0.72, 0.68, 0.60), "opaque", 0.55, 0.0),     # 0 floor
    ("brick", (0.70, 0.62, 0.52), "opaque", 0.85, 0.0),    # 1 walls
    ("noise", (0.78, 0.76, 0.70), "opaque", 0.70, 0.0),    # 2 columns
    ("noise", (0.66, 0.64, 0.60), "opaque", 0.80, 0.0),    # 3 arches
    ("brick", (0.60, 0.55, 0.50), "opaque", 0.90, 0.0),    # 4 ceiling
    ("fabric", (0.75, 0.12, 0.10), "opaque", 0.95, 0.0),   # 5 curtain red
    ("fabric", (0.10, 0.30, 0.70), "opaque", 0.95, 0.0),   # 6 curtain blue
    ("fabric", (0.12, 0.55, 0.20), "opaque", 0.95, 0.0),   # 7 curtain green
    ("noise", (0.20, 0.55, 0.15), "cutout", 0.80, 0.0),    # 8 foliage (alpha cut-outs)
    ("noise", (0.75, 0.85, 0.90), "half", 0.10, 0.0),      # 9 glass (alpha 128/255)
    ("noise", (0.85, 0.65, 0.25), "opaque", 0.30, 1.0),    # 10 lion (brass)
    ("tile", (0.55, 0.50, 0.45), "opaque", 0.60, 0.0),     # 11 balcony
    ("noise", (0.45, 0.30, 0.20), "opaque", 0.75, 0.0),    # 12 vases
    ("fabric", (0.80, 0.70, 0.20), "cutout", 0.90, 0.0),   # 13 lace banner (alpha cut-outs)
    ("noise", (0.50, 0.50, 0.55), "opaque", 0.35, 1.0),    # 14 chains (steel)
]


def sponza_standin(detail: float = 1.0, tex_size: int = 1024, n_materials: int = 25, seed: int = SEED) -> Scene:
    """Atrium of about 3000 x 1100 x 1400 units: floor, walls and ceiling, two tiers of
    arcaded columns along the nave, balconies, hanging curtains, foliage and lace with
    alpha cut-outs, a half-transparent pane, vases and a dense 'lion head' relief.
    detail = 1.0 gives about 262k triangles; tests use small values.  The instance is the
    reference's Sponza instance (translation (60,0,35), state.rs:407-412), so local
    coordinates are shifted back by it to keep the fog box centred on the floor."""
    from . import abi

    rng = np.random.default_rng(seed)
    k = lambda base: max(2, int(round(base * detail * 0.9)))  # tessellation along one axis
    kc = lambda base: max(1, int(round(base * np.sqrt(detail))))  # object counts
    off = np.array([-60.0, 0.0, -35.0])  # undo the instance translation

    mats_parts = {i: [] for i in range(n_materials)}

    def add(mat, part):
        v, i = part
        v = v.copy()
        v[:, 0:3] += off.astype(np.float32)
        mats_parts[mat % n_materials].append((v, i))

    X0, X1, Z0, Z1, YH = -1500.0, 1500.0, -700.0, 700.0, 1100.0
    # floor (+y), ceiling (-y), four walls (inward)
    add(0, _patch((X0, 0, Z0), (0, 0, Z1 - Z0), (X1 - X0, 0, 0), k(48), k(96), (7.0, 15.0)))
    add(4, _patch((X0, YH, Z0), (X1 - X0, 0, 0), (0, 0, Z1 - Z0), k(48), k(24), (15.0, 7.0)))
    add(1, _patch((X0, 0, Z1), (X1 - X0, 0, 0), (0, YH, 0), k(96), k(36), (12.0, 4.4)))
    add(1, _patch((X1, 0, Z0), (X0 - X1, 0, 0), (0, YH, 0), k(96), k(36), (12.0, 4.4)))
    add(1, _patch((X0, 0, Z0), (0, 0, Z1 - Z0), (0, YH, 0), k(48), k(36), (5.6, 4.4)))
    add(1, _patch((X1, 0, Z1), (0, 0, Z0 - Z1), (0, YH, 0), k(48), k(36), (5.6, 4.4)))

    # two tiers of columns along the nave at z = +-260
    ncol = kc(12)
    xs = np.linspace(-1250.0, 1250.0, ncol)
    for tier, (y0, h, r, mat) in enumerate(((0.0, 400.0, 42.0, 2), (450.0, 330.0, 30.0, 2))):
        for zc in (-260.0, 260.0):
            for xc in xs:
                def cyl(S, T, xc=xc, zc=zc, y0=y0, h=h, r=r):
                    th = -S * 2 * np.pi
                    rr = r * (1.0 + 0.06 * np.cos(T * np.pi * 2) + 0.03 * np.cos(th * 12))
                    return np.stack([xc + rr * np.cos(th), y0 + T * h, zc + rr * np.sin(th)], axis=2)
                add(mat, _surface(cyl, k(40), k(20), (3.0, 5.0)))
                add(3, _box((xc - 60, y0 + h, zc - 60), (xc + 60, y0 + h + 25, zc + 60), (k(3), 1, k(3))))
    # arches between ground-tier columns
    for zc in (-260.0, 260.0):
        for xa, xb in zip(xs[:-1], xs[1:]):
            def arch(S, T, xa=xa, xb=xb, zc=zc):
                th = np.pi * (1.0 - S)
                rad = 0.5 * (xb - xa) - 45.0
                return np.stack([0.5 * (xa + xb) + rad * np.cos(th), 300.0 + 0.4 * rad * np.sin(th) + 0 * T, zc - 45.0 + 90.0 * T], axis=2)
            add(3, _surface(arch, k(24), k(4), (4.0, 1.0), flip=True))
    # balcony slabs and railings
    for zc0, zc1 in ((-700.0, -215.0), (215.0, 700.0)):
        add(11, _box((X0, 425.0, zc0), (X1, 450.0, zc1), (k(60), 1, k(10)), 10.0))
    for zc in (-215.0, 215.0):
        add(14, _box((X0, 450.0, zc - 4), (X1, 520.0, zc + 4), (k(120), k(3), 1), 20.0))
    # curtains (wavy, two-sided) between upper columns
    cols = [5, 6, 7]
    ncur = kc(8)
    for ci, xc in enumerate(np.linspace(-1100.0, 1100.0, ncur)):
        for zc in (-262.0, 262.0):
            def cur(S, T, xc=xc, zc=zc, ci=ci):
                return np.stack([xc - 90 + 180 * S, 780.0 - 320.0 * T, zc + 14.0 * np.sin(S * 9.0 + ci) * (0.3 + T)], axis=2)
            add(cols[ci % 3], _surface(cur, k(48), k(48), (2.0, 3.0)))
            add(cols[ci % 3], _surface(cur, k(48), k(48), (2.0, 3.0), flip=True))
    # lace banners with cut-out alpha across the nave
    nban = kc(4)
    for bi, xc in enumerate(np.linspace(-700.0, 900.0, nban)):
        def ban(S, T, xc=xc, bi=bi):
            return np.stack([xc + 10.0 * np.sin(T * 5 + bi), 700.0 - 260.0 * T, -150.0 + 300.0 * S], axis=2)
        add(13, _surface(ban, k(24), k(24), (3.0, 3.0)))
        add(13, _surface(ban, k(24), k(24), (3.0, 3.0), flip=True))
    # foliage: vases with crossed cut-out quads
    nveg = kc(10)
    for vi, xc in enumerate(np.linspace(-1150.0, 1150.0, nveg)):
        for zc in (-150.0, 150.0):
            def vase(S, T, xc=xc, zc=zc):
                th = -S * 2 * np.pi
                rr = 18.0 + 16.0 * np.sin(T * np.pi) ** 2
                return np.stack([xc + rr * np.cos(th), 70.0 * T, zc + rr * np.sin(th)], axis=2)
            add(12, _surface(vase, k(24), k(12), (2.0, 1.0)))
            for ang in (0.0, np.pi / 3, 2 * np.pi / 3):
                dx, dz = 32.0 * np.cos(ang), 32.0 * np.sin(ang)
                for sgn in (1.0, -1.0):
                    add(8, _patch((xc - sgn * dx, 60.0, zc - sgn * dz), (2 * sgn * dx, 0, 2 * sgn * dz), (0, 85.0, 0), k(4), k(4), (1.0, 1.0)))
    # half-transparent pane across the nave, seen by the reference camera
    add(9, _patch((350.0, 60.0, 100.0), (0, 0, -200.0), (0, 180.0, 0), k(6), k(6), (1.0, 1.0)))
    add(9, _patch((350.0, 60.0, -100.0), (0, 0, 200.0), (0, 180.0, 0), k(6), k(6), (1.0, 1.0)))
    # lion-head relief on the far wall: dense bumpy dome
    def lion(S, T):
        th = S * np.pi
        ph = T * np.pi
        rr = 170.0 * (1.0 + 0.10 * np.sin(9 * th) * np.sin(7 * ph) + 0.05 * np.cos(23 * th + 11 * ph))
        return np.stack([1495.0 - rr * np.sin(th) * np.sin(ph) * 0.6, 420.0 + rr * np.cos(ph), rr * np.cos(th) * np.sin(ph)], axis=2)
    add(10, _surface(lion, k(150), k(110), (2.0, 2.0)))

    # materials / textures
    textures: List[Tuple[np.ndarray, bool]] = []
    materials: List[Material] = []
    for mi in range(n_materials):
        kind, color, amode, rough, metal = _PALETTE[mi % len(_PALETTE)]
        if mi >= len(_PALETTE):
            color = tuple(np.clip(np.asarray(color) * (0.6 + 0.4 * rng.random(3)), 0, 1))
        base, normal, mr = _material_textures(rng, tex_size, kind, color, amode, rough, metal)
        t0 = len(textures)
        textures += [(base, True), (normal, False), (mr, False)]
        materials.append(Material(t0, t0 + 1, t0 + 2, 1.0, 1.0))
    # spread the extra materials (beyond the palette) over wall/column segments so that
    # every material is referenced by some primitive
    meshes: List[Mesh] = []
    for mi in range(n_materials):
        parts = mats_parts[mi]
        if not parts:
            # a small plaque on the wall keeps the material in use
            j = mi - len(_PALETTE)
            plaque = _patch((-1300.0 + 110.0 * j, 150.0, Z1 - 2.0), (100.0, 0, 0), (0, 100.0, 0), k(4), k(4), (1.0, 1.0))
            v = plaque[0].copy()
            v[:, 0:3] += off.astype(np.float32)
            parts = [(v, plaque[1])]
        # one glTF primitive per material and per ~65k vertices, like an exporter would split
        v, i = _merge(parts)
        meshes.append(Mesh(v, i, mi))
    geometry = Model(meshes, materials)
    cube = cube_model()
    return Scene(
        textures=textures, geometry=geometry, fog=cube, marker=cube_model(),
        geometry_instance=abi.instance_raw(REFERENCE_GEOMETRY_INSTANCE["position"], (1, 0, 0, 0), REFERENCE_GEOMETRY_INSTANCE["scale"]),
        fog_instance=abi.instance_raw(REFERENCE_FOG_INSTANCE["position"], (1, 0, 0, 0), REFERENCE_FOG_INSTANCE["scale"]),
    )


def cube_scene(textures: Optional[List[Tuple[np.ndarray, bool]]] = None, tex_size: int = 64) -> Scene:
    """The reference's smallest scene: Cube.glb as the geometry model, instance scale 100 at the origin.
    Without `textures` (base colour, metallic-roughness as decoded from the GLB) a small
    procedural pair is used.  Cube.glb has no normal texture, so the normal slot falls back
    to image 0 read as linear (resources.rs:74,89)."""
    from . import abi

    cube = cube_model()
    if textures is None:
        rng = np.random.default_rng(SEED)
        base, _, _ = _material_textures(rng, tex_size, "tile", (0.8, 0.7, 0.6), "opaque", 0.5, 0.0)
        mr = np.zeros((tex_size, tex_size, 4), np.uint8)
        mr[..., 1] = 20
        mr[..., 3] = 255
        img0, img1 = base, mr
    else:
        img0, img1 = textures[0], textures[1]
    tex = [(img0, True), (img0, False), (img1, False)]
    geometry = Model([Mesh(cube.meshes[0].vertices, cube.meshes[0].indices, 0)], [Material(0, 1, 2, 1.0, 1.0)])
    return Scene(
        textures=tex, geometry=geometry, fog=cube_model(), marker=cube_model(),
        geometry_instance=abi.instance_raw((0.0, 0.0, 0.0), (1, 0, 0, 0), (100.0, 100.0, 100.0)),
        fog_instance=abi.instance_raw(REFERENCE_FOG_INSTANCE["position"], (1, 0, 0, 0), REFERENCE_FOG_INSTANCE["scale"]),
    )


def flythrough_pose(k: int, n: int = 512):
    """Pose k of the n-pose flythrough used by the frame-parallel configuration: the camera
    walks an ellipse around the nave at eye height and looks along its direction of travel.
    Returns (position, pitch_deg, yaw_deg)."""
    a = 2.0 * np.pi * k / n
    pos = (-700.0 * np.cos(a), 150.0 + 40.0 * np.sin(2 * a), 150.0 * np.sin(a))
    dx, dz = 700.0 * np.sin(a), 150.0 * np.cos(a)
    yaw = float(np.degrees(np.arctan2(dz, dx)))
    return pos, 0.0, yaw
