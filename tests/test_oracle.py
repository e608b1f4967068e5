"""CPU-only tests of the oracle (the CPU restatement) and of the product's arithmetic
against it.

The reference holds no golden vectors for this path (no tests at all), so the oracle's
parity with the reference is unpinned; what is checked here is
  * the oracle against independent float64 numpy restatements of the WGSL formulas and
    against analytic properties (so that a slip in the C++ is caught),
  * the oracle against the committed fixtures in tests/golden/ (regression pin),
  * the product's __host__ __device__ arithmetic (csrc/wr_device.h, built for the host in
    tests/hostcheck) bit for bit against the oracle.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import wgpu_renderer_amd  # noqa: F401
from wgpu_renderer_amd import abi, scene
from golden_inputs import GOLDEN, cube_config1, standin_small, unit_inputs
from oracle_lib import Oracle, f3, lib

L = lib()
HC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck", "libwr_hostcheck.so")


def f32(x):
    return np.float32(x)


def from_bits(b):
    return np.uint32(b).view(np.float32)


def bits(x):
    return int(np.float32(x).view(np.uint32))


# ------------------------------------------------------------------ unit formulas
def cephes_sinf_numpy(x):
    """Rule R4 restated with numpy float32 scalars (no shared code with the C++): the published
    single-precision Cephes / sse_mathfun sine.  Fused steps are done in float64 and rounded
    once, which equals an fma whenever the float64 product-sum is exact enough; the test below
    only uses it where that holds (|x| small enough that y * DP fits 53 bits)."""
    f = np.float32
    sign = x < 0
    ax = f(abs(x))
    y = f(ax * f(1.27323954473516))
    j = (int(y) + 1) & ~1
    y = f(j)
    if j & 4:
        sign = not sign
    fma = lambda a, b, c: f(np.float64(a) * np.float64(b) + np.float64(c))
    r = fma(y, f(-0.78515625), ax)
    r = fma(y, f(-2.4187564849853515625e-4), r)
    r = fma(y, f(-3.77489497744594108e-8), r)
    z = f(r * r)
    if j & 2 == 0:
        p = fma(f(-1.9515295891e-4), z, f(8.3321608736e-3))
        p = fma(p, z, f(-1.6666654611e-1))
        p = f(p * z)
        res = fma(p, r, r)
    else:
        p = fma(f(2.443315711809948e-5), z, f(-1.388731625493765e-3))
        p = fma(p, z, f(4.166664568298827e-2))
        p = f(f(p * z) * z)
        p = fma(z, f(-0.5), p)
        res = f(p + f(1.0))
    return f(-res) if sign else f(res)


def test_sine_accuracy_and_symmetry():
    xs = np.linspace(-20, 20, 4001, dtype=np.float32)
    got = np.array([L.wro_sin(float(x)) for x in xs])
    assert np.abs(got - np.sin(xs.astype(np.float64))).max() < 1.2e-7
    # the numpy restatement of the published algorithm agrees bit for bit
    rng = np.random.default_rng(9)
    for x in np.concatenate([xs[::40], rng.uniform(-3e4, 3e4, 400).astype(np.float32)]):
        assert bits(L.wro_sin(float(x))) == bits(cephes_sinf_numpy(np.float32(x))), float(x)
    # arguments the noise hash reaches (lattice dot products up to millions): the three-part
    # reduction keeps the error at a few 1e-7 (exhaustive maximum over [0, 2^23): 1.77e-7)
    big = np.concatenate([rng.uniform(-8388607.0, 8388607.0, 200000), np.arange(1, 100001) * np.pi]).astype(np.float32)
    out = np.empty_like(big)
    fp = C.POINTER(C.c_float)
    L.wro_sin_many.argtypes = [fp, C.c_int, fp]
    L.wro_sin_many(big.ctypes.data_as(fp), len(big), out.ctypes.data_as(fp))
    assert np.abs(out.astype(np.float64) - np.sin(big.astype(np.float64))).max() < 1.8e-7
    assert np.abs(out).max() <= 1.0
    for x in (0.3, 2.5, 1000.25, 123456.7, 8.0e6, 9.0e8):
        assert L.wro_sin(-x) == -L.wro_sin(x)
    # beyond 2^24 the three-part reduction runs out of bits, as in the published algorithm: the
    # value is finite and deterministic but no longer a sine; from 2^30 on the rule says 0
    assert abs(L.wro_sin(4.0e6)) <= 1.0 and np.isfinite(L.wro_sin(1.0e9)) and L.wro_sin(1.0e9) != 0.0
    assert L.wro_sin(1073741824.0) == 0.0 and L.wro_sin(2.0e9) == 0.0 and L.wro_sin(float("inf")) == 0.0
    assert not np.isnan(L.wro_sin(float("nan")))


def test_hash_fraction_never_rounds_up_to_one():
    """The device takes the hash's fractional part with the hardware's fraction instruction,
    which differs from s - floor(s) only where that rounds up to 1.0.  Every float argument
    with 2^-20 <= |d| < 2^30 (a lattice dot product is 0 or at least 2^-20 in magnitude) is
    tried here: no such case exists."""
    from concurrent.futures import ThreadPoolExecutor
    L.wro_hash_fract_exceptions.restype = C.c_uint64
    L.wro_hash_fract_exceptions.argtypes = [C.c_uint32] * 4
    lo, hi = bits(2.0 ** -20), bits(2.0 ** 30)
    with ThreadPoolExecutor(8) as ex:
        bad = sum(ex.map(lambda t: L.wro_hash_fract_exceptions(lo, hi, 8, t), range(8)))
    assert bad == 0
    assert L.wro_hash_fract_exceptions(bits(0.0), bits(0.0) + 1, 1, 0) == 0


def test_rand_is_the_sin_hash():
    # noise.wgsl:3-5 with the restatement's f32 evaluation order
    rng = np.random.default_rng(1)
    for _ in range(200):
        p = np.floor(rng.uniform(-2000, 2000, 3)).astype(np.float32)
        d = f32(f32(f32(p[0] * f32(12.345)) + f32(p[1] * f32(67.89))) + f32(p[2] * f32(412.12)))
        s = f32(f32(L.wro_sin(float(d))) * f32(42123.45))
        expect = f32(f32(s - np.floor(s)) * f32(2.0)) - f32(1.0)
        got = L.wro_rand(f3(p))
        assert got == expect and -1.0 <= got < 1.0


def test_value_noise_interpolates_lattice_hashes():
    rng = np.random.default_rng(2)
    for _ in range(50):
        u = np.floor(rng.uniform(-500, 500, 3)).astype(np.float32)
        assert L.wro_value_noise(f3(u)) == L.wro_rand(f3(u))  # weights (0,0,0) pick corner a
        # halfway along x between two lattice points: smoothstep(0.5) = 0.5
        a, b = L.wro_rand(f3(u)), L.wro_rand(f3(u + np.array([1, 0, 0], np.float32)))
        mid = L.wro_value_noise(f3(u + np.array([0.5, 0, 0], np.float32)))
        assert np.isclose(mid, 0.5 * a + 0.5 * b, atol=1e-6)


def test_fbm_and_fog_noise_ranges():
    rng = np.random.default_rng(3)
    vals = []
    for _ in range(300):
        p = rng.uniform(-1500, 1500, 3).astype(np.float32)
        f = L.wro_fbm(f3(p * 0.01))
        n = L.wro_fog_noise(f3(p), 2.0)
        assert 0.0 <= f <= 1.0 and 0.05 - 1e-6 <= n <= 1.0 + 1e-6  # fog.wgsl:61
        vals.append(n)
    assert np.std(vals) > 0.02  # it is noise, not a constant


def test_depth_to_linear_keeps_the_reference_quirk():
    # fog.wgsl:118-123 assumes a [0,1] projection although cgmath::perspective is GL-style:
    # at depth 0 it returns near, at depth 1 far, and it is NOT the eye distance in between.
    assert L.wro_depth_to_linear(0.0, 1.0, 3000.0) == 1.0
    assert np.isclose(L.wro_depth_to_linear(1.0, 1.0, 3000.0), 3000.0, rtol=1e-4)
    d = 0.99
    assert np.isclose(L.wro_depth_to_linear(d, 1.0, 3000.0), 3000.0 / (3000.0 + d * (1.0 - 3000.0)), rtol=1e-4)


def test_srgb_tables_round_trip_and_monotone():
    dec = np.array([L.wro_srgb_decode(c) for c in range(256)], np.float64)
    assert dec[0] == 0.0 and np.isclose(dec[255], 1.0) and np.all(np.diff(dec) > 0)
    s = np.arange(256) / 255.0
    ref = np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)
    assert np.allclose(dec, ref, rtol=1e-6)
    for c in range(256):
        assert L.wro_srgb_encode(float(np.float32(dec[c]))) == c  # decode -> encode is the identity
    xs = np.linspace(0, 1, 5001)
    enc = np.array([L.wro_srgb_encode(float(x)) for x in xs])
    assert np.all(np.diff(enc) >= 0)
    exact = np.where(xs <= 0.0031308, xs * 12.92, 1.055 * xs ** (1 / 2.4) - 0.055) * 255
    assert np.abs(enc - exact).max() <= 0.5 + 1e-3  # nearest code
    assert L.wro_srgb_encode(float("nan")) == 0 and L.wro_srgb_encode(-1.0) == 0 and L.wro_srgb_encode(7.0) == 255


def brdf_np(n, l, v, h, albedo, rough, metal):
    """brdf.wgsl in float64, including its quirks: PI = 3.14159, a = roughness (not squared),
    k = roughness."""
    PI = 3.14159
    f0 = 0.04 * (1 - metal) + albedo * metal
    fres = f0 + (1 - f0) * (1 - max(h @ v, 0.0)) ** 5
    a2 = rough * rough
    ndh = max(n @ h, 0.0)
    den = ndh * ndh * (a2 - 1) + 1
    ndf = a2 / (PI * den * den)
    ndv, ndl = max(n @ v, 0.0), max(n @ l, 0.0)
    g = (ndv / (ndv * (1 - rough) + rough)) * (ndl / (ndl * (1 - rough) + rough))
    spec = ndf * g * fres / (4 * ndv * ndl + 0.0001)
    kd = (1 - fres) * (1 - metal)
    return (kd * albedo / PI + spec) * ndl


def test_brdf_and_ambient_against_float64_restatement():
    for b in unit_inputs()["brdf"]:
        b64 = b.astype(np.float64)
        out = (C.c_float * 3)()
        L.wro_brdf(f3(b[0:3]), f3(b[3:6]), f3(b[6:9]), f3(b[9:12]), f3(b[12:15]), float(b[15]), float(b[16]), out)
        ref = brdf_np(b64[0:3], b64[3:6], b64[6:9], b64[9:12], b64[12:15], b64[15], b64[16])
        assert np.allclose(np.array(out[:]), ref, rtol=2e-4, atol=1e-7)
    for a in unit_inputs()["ambient"]:
        a64 = a.astype(np.float64)
        out = (C.c_float * 3)()
        L.wro_ambient((C.c_float * 4)(*a[0:4]), float(a[4]), float(a[5]), out)
        att = 1.0 / (1.0 - 0.75 * a64[4] + 0.25 * a64[4] ** 2)  # light.wgsl:68-70
        ref = 0.01 + max(a64[5], 0.0) * 0.03 * a64[0:3] * a64[3] * att
        assert np.allclose(np.array(out[:]), ref, rtol=1e-5)


# ------------------------------------------------------------------ golden fixtures
def test_oracle_matches_committed_vectors():
    vec = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))
    for xb, yb in vec["sin"]:
        assert bits(L.wro_sin(float(from_bits(xb)))) == yb
    for pb, yb in vec["value_noise"]:
        assert bits(L.wro_value_noise(f3([from_bits(b) for b in pb]))) == yb
    for pb, yb in vec["fog_noise"]:
        assert bits(L.wro_fog_noise(f3([from_bits(b) for b in pb]), 2.0)) == yb
    for xb, c in vec["srgb_encode"]:
        assert L.wro_srgb_encode(float(from_bits(xb))) == c
    for clipb, shadow, words in vec["setup"]:
        out = (C.c_uint32 * (28 * 8))()
        n = L.wro_setup_triangle((C.c_float * 12)(*[from_bits(b) for b in clipb]), 640, 360, shadow, out, 8)
        assert list(out[: 28 * n]) == words


@pytest.mark.parametrize("name,maker", [("cube_frame_160x90", lambda: cube_config1(160, 90)),
                                        ("standin_frame_192x108", lambda: standin_small(192, 108))])
def test_oracle_frames_match_committed_fixtures(name, maker):
    sc, cam, light, glob, passes, w, h, s = maker()
    o = Oracle(w, h, s)
    o.upload_scene(sc)
    o.render_frame(cam, light, glob, passes, threads=3)  # band count must not change results
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert np.array_equal(o.read_visibility(), g["visibility"])
    assert np.array_equal(o.read_depth(), g["depth"])
    assert np.array_equal(o.read_color(), g["color"])
    if passes & abi.PASS_SHADOWS:
        assert np.array_equal(o.read_shadow_map()[3], g["shadow_face3"])
    o.close()


# ------------------------------------------------------------------ rendered-image properties
def test_cube_config1_image_properties():
    sc, cam, light, glob, passes, w, h, s = cube_config1(160, 90)
    o = Oracle(w, h, s)
    o.upload_scene(sc)
    o.render_frame(cam, light, glob, passes, threads=2)
    vis, depth, col = o.read_visibility(), o.read_depth(), o.read_color()
    covered = vis != 0xFFFFFFFF
    assert 500 < covered.sum() < 5000                 # the cube is in view, not filling it
    assert np.all(depth[~covered] == 1.0) and np.all(depth[covered] < 1.0)
    assert np.all(col[~covered] == [0, 0, 0, 255])    # clear colour (0,0,0,1), state.rs:696-701
    assert np.all(col[..., 3] == 255)                 # opaque texels keep alpha 1
    assert (vis[covered] >> 3).max() < 12 + 12        # 12 cube triangles (+ marker ids)
    # back-face culling: from this camera at most 3 faces (6 triangles) are visible
    assert len(np.unique(vis[covered] >> 3)) <= 6 + 12
    # t = 0 gives a black light: only the 0.01 ambient term remains (state.rs:615-617)
    o.render_frame(cam, abi.light_at_time(0.0), abi.global_uniforms(0.0, 0), passes, threads=2)
    assert o.read_color()[..., :3].max() <= 30
    o.close()


def test_fog_needs_camera_outside_the_volume_and_shadows_darken():
    sc, cam, light, glob, passes, w, h, s = standin_small(128, 72)
    o = Oracle(w, h, s)
    o.upload_scene(sc)
    o.render_frame(cam, light, glob, abi.PASS_ALL, threads=4)
    with_fog = o.read_color().astype(np.int32)
    c_fog = o.counters()
    o.render_frame(cam, light, glob, abi.PASS_SHADOWS | abi.PASS_LIGHT_DEBUG, threads=4)
    no_fog = o.read_color().astype(np.int32)
    assert c_fog["fog_fragments"] > 500 and (with_fog != no_fog).any()
    # camera inside the fog box: its front faces are culled, the pass draws nothing
    inside = abi.camera_uniform((-500.0, 30.0, 0.0), 0.0, 0.0, 55.0, w, h)
    o.render_frame(inside, light, glob, abi.PASS_ALL, threads=4)
    assert o.counters()["fog_fragments"] == 0
    # shadow maps only ever remove direct light
    o.render_frame(cam, light, abi.global_uniforms(2.0, 0), abi.PASS_LIGHT_DEBUG, threads=4)
    lit = o.read_color().astype(np.int32)
    assert no_fog[..., :3].sum() < lit[..., :3].sum()
    # fully lit PCF value is 16/25 (light.wgsl:13-26), so even unshadowed pixels are darker
    assert (no_fog[..., :3] <= lit[..., :3] + 1).mean() > 0.99
    o.close()


# ------------------------------------------------------------------ product arithmetic == oracle
@pytest.fixture(scope="module")
def hc():
    if not os.path.exists(HC_PATH):
        pytest.skip("tests/hostcheck/libwr_hostcheck.so not built (run __graft_entry__.build())")
    H = C.CDLL(HC_PATH)
    H.hc_sin.restype = C.c_float
    H.hc_sin.argtypes = [C.c_float]
    for n in ("hc_value_noise", "hc_fbm"):
        getattr(H, n).restype = C.c_float
        getattr(H, n).argtypes = [C.POINTER(C.c_float)]
    H.hc_fog_noise.restype = C.c_float
    H.hc_fog_noise.argtypes = [C.POINTER(C.c_float), C.c_float]
    H.hc_depth_to_linear.restype = C.c_float
    H.hc_depth_to_linear.argtypes = [C.c_float] * 3
    H.hc_brdf.argtypes = [C.POINTER(C.c_float)] * 5 + [C.c_float, C.c_float, C.POINTER(C.c_float)]
    H.hc_ambient.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]
    H.hc_srgb_encode.restype = C.c_uint
    H.hc_srgb_encode.argtypes = [C.c_float]
    H.hc_setup_triangle.restype = C.c_int
    H.hc_setup_triangle.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.c_int]
    return H


def test_product_arithmetic_is_bit_identical_to_oracle(hc):
    ui = unit_inputs()
    rng = np.random.default_rng(11)
    for x in np.concatenate([ui["sin"], rng.uniform(-5e6, 5e6, 3000).astype(np.float32)]):
        assert bits(hc.hc_sin(float(x))) == bits(L.wro_sin(float(x)))
    for p in np.concatenate([ui["points"], rng.uniform(-3000, 3000, (500, 3)).astype(np.float32)]):
        assert hc.hc_value_noise(f3(p)) == L.wro_value_noise(f3(p))
        assert hc.hc_fbm(f3(p * 0.05)) == L.wro_fbm(f3(p * 0.05))
        assert hc.hc_fog_noise(f3(p), 7.5) == L.wro_fog_noise(f3(p), 7.5)
    for b in ui["brdf"]:
        a, o = (C.c_float * 3)(), (C.c_float * 3)()
        args = (f3(b[0:3]), f3(b[3:6]), f3(b[6:9]), f3(b[9:12]), f3(b[12:15]), float(b[15]), float(b[16]))
        hc.hc_brdf(*args, a)
        L.wro_brdf(*args, o)
        assert a[:] == o[:]
    for x in np.concatenate([ui["linear"], rng.uniform(0, 1, 2000).astype(np.float32)]):
        assert hc.hc_srgb_encode(float(x)) == L.wro_srgb_encode(float(x))
    for d in ui["depth"]:
        assert hc.hc_depth_to_linear(float(d), 1.0, 3000.0) == L.wro_depth_to_linear(float(d), 1.0, 3000.0)


def test_product_fast_sine_forms_equal_the_guarded_form_on_every_float_of_their_range(hc):
    """The hot path uses the form of the sine whose octant arithmetic stays in floating point,
    allowed for |x| <= 13176794; every float of that range (both signs) is compared here with the
    guarded form, which the tests above compare with the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    hc.hc_sin_fast_mismatches.restype = C.c_ulonglong
    hc.hc_sin_fast_mismatches.argtypes = [C.c_uint] * 4
    hc.hc_sin_fast_range.restype = C.c_float
    hi = bits(hc.hc_sin_fast_range()) + 1
    with ThreadPoolExecutor(8) as ex:
        bad = sum(ex.map(lambda t: hc.hc_sin_fast_mismatches(0, hi, 8, t), range(8)))
    assert bad == 0


def test_product_triangle_setup_is_bit_identical_to_oracle(hc):
    """Random triangles around a perspective camera, many crossing the near plane or the
    guard band: same sub-triangles, same snapped vertices, depth planes, clip weights and
    (for shadow records) depth bias."""
    rng = np.random.default_rng(5)
    f, n = 3000.0, 1.0
    fl = 1 / np.tan(np.radians(55) / 2)
    total = 0
    for i in range(6000):
        P = rng.uniform(-1, 1, (3, 3)) * 3 + np.array([0, 0, rng.uniform(-4, 2)])
        clip = []
        for k in range(3):
            x, y, z = P[k] * rng.choice([1, 1, 1, 50])
            clip += [fl / 1.777 * x, fl * y, (f + n) / (n - f) * z + 2 * f * n / (n - f), -z]
        cl = (C.c_float * 12)(*np.array(clip, np.float32).tolist())
        for shadow in (0, 1):
            oa, ob = (C.c_uint32 * (28 * 8))(), (C.c_uint32 * (28 * 8))()
            na = L.wro_setup_triangle(cl, 640, 360, shadow, oa, 8)
            nb = hc.hc_setup_triangle(cl, 640, 360, shadow, ob, 8)
            assert na == nb
            A = np.array(oa[: 28 * na], np.uint32).reshape(na, 28)
            B = np.array(ob[: 28 * nb], np.uint32).reshape(nb, 28)
            if shadow:
                A[:, 10] = 0  # shadow records do not keep 1/w of vertex 0 (the bias takes its slot)
            assert np.array_equal(A, B)
            total += na
    assert total > 2000


def test_product_noise_through_tables_equals_the_direct_evaluation(hc):
    """Host build of the table lookup (value_noise_tab: position in the box in f32, box test on
    the bits of u - lo, index by fused multiply-adds): fog_noise through tables built around the
    point equals the direct evaluation and the oracle bit for bit, also where a coordinate sits on
    a lattice plane (every octave's floor is then the coordinate itself), at zero and minus zero,
    and at negative coordinates."""
    hc.hc_fog_noise_tab.restype = C.c_float
    hc.hc_fog_noise_tab.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(31)
    pts = [rng.uniform(-1500, 1500, 3) for _ in range(300)]
    pts += [np.array(v, np.float64) for v in ([0, 0, 0], [-0.0, -0.0, -0.0], [100, 200, -300], [-100, 20, 40], [12.5, -6.25, 3.125],
                                              [1e-30, -1e-30, 0.0], [-700, 50, 599.99], [-0.78125, 0.390625, -0.1953125])]
    hits = 0
    for time in (0.0, 2.0, 7.5):
        for p in pts:
            levels = C.c_int(0)
            got = hc.hc_fog_noise_tab(f3(np.asarray(p, np.float32)), time, 4096, C.byref(levels))
            want = L.wro_fog_noise(f3(np.asarray(p, np.float32)), time)
            assert bits(got) == bits(want), (p, time)
            hits += levels.value
    assert hits > 8 * len(pts)  # most levels of most points were read from tables
