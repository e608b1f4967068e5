"""How far the picture moves when one of the oracle's single-sourced rules is decided the other
plausible way.  CPU only.

The oracle fixes what the reference leaves to wgpu / naga / the Vulkan driver by explicit rules
(header of oracle/wr_oracle.cpp).  Some of them have no second source in this repository: the
order of operations inside `rand`'s dot product (R1), the form of `mix` (R3), the fused steps
of the sine (R4), coarse or fine quad derivatives (R12), which filter a level of detail of
exactly zero selects (R13), how a colour is rounded to its 8-bit code (R15).  For each, this
script builds the oracle with the alternative compiled in (oracle/_variants/, not committed),
renders the small stand-in frame (tests/golden_inputs.standin_small: shadows + fog, 192x108),
a larger one with 256^2 textures and the Cube frame, evaluates fog_noise at 4096 points, and
records the distance from the default build's results:

    tests/golden/rule_sensitivity.json

so that a later comparison with a real wgpu-on-lavapipe render can be triaged by the size and
the place of its differences.  No rule changes because of these numbers.

    python scripts/rule_sensitivity.py            # rebuild the variants and rewrite the JSON
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ORACLE = os.path.join(ROOT, "oracle")
VARIANTS = os.path.join(ORACLE, "_variants")
OUT = os.path.join(ROOT, "tests", "golden", "rule_sensitivity.json")

# name -> (macro, rule, what the alternative is)
ALTERNATIVES = {
    "R1_fused_dot": ("WRO_ALT_R1_FUSED_DOT", "R1", "rand's dot product as fma(z, 412.12, fma(x, 12.345, y*67.89)) instead of three rounded products and two rounded sums"),
    "R3_mix_one_mul": ("WRO_ALT_R3_MIX_ONE_MUL", "R3", "mix(a, b, t) = a + t*(b - a) instead of a*(1 - t) + b*t"),
    "R4_unfused_sine": ("WRO_ALT_R4_UNFUSED_SINE", "R4", "the Cephes sine with every multiply and add rounded on its own instead of fused multiply-adds"),
    "R12_fine_derivatives": ("WRO_ALT_R12_FINE_DERIVATIVES", "R12", "ddx along the fragment's own row and ddy along its own column instead of the quad's top-left differences"),
    "R13_tie_minifies": ("WRO_ALT_R13_TIE_MINIFIES", "R13", "a level of detail of exactly zero (rho == 1) selects the minification filter (nearest) instead of the magnification filter"),
    "R15_truncate": ("WRO_ALT_R15_TRUNCATE", "R15", "colour and alpha codes rounded down instead of to nearest"),
}
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mfma", "-fno-fast-math", "-pthread", "-shared"]


def build_variant(name):
    os.makedirs(VARIANTS, exist_ok=True)
    out = os.path.join(VARIANTS, f"libwr_oracle_{name}.so")
    src = os.path.join(ORACLE, "wr_oracle.cpp")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["g++", *CXXFLAGS, "-D" + ALTERNATIVES[name][0], "-o", out, src], check=True)
    return out


def render(library, maker):
    from oracle_lib import Oracle
    sc, cam, light, glob, passes, w, h, s = maker()
    o = Oracle(w, h, s, library=library)
    o.upload_scene(sc)
    o.render_frame(cam, light, glob, passes, threads=8)
    out = dict(vis=o.read_visibility(), color=o.read_color(), depth=o.read_depth())
    o.close()
    return out


def fog_mask(maker):
    """Pixels the fog pass changes: the same frame with and without WR_PASS_FOG."""
    from oracle_lib import Oracle
    from wgpu_renderer_amd import abi
    sc, cam, light, glob, passes, w, h, s = maker()
    if not passes & abi.PASS_FOG:
        return None
    frames = []
    for p in (passes, passes & ~abi.PASS_FOG):
        o = Oracle(w, h, s)
        o.upload_scene(sc)
        o.render_frame(cam, light, glob, p, threads=8)
        frames.append(o.read_color())
        o.close()
    return (frames[0] != frames[1]).any(axis=2)


def distance(base, alt, fog):
    from parity import compare
    m = compare(base["vis"], alt["vis"], base["color"], alt["color"])
    diff = np.abs(base["color"].astype(np.int32) - alt["color"].astype(np.int32)).max(axis=2)
    out = {
        "id_match": m["id_match"], "psnr_db": None if m["psnr"] == float("inf") else round(m["psnr"], 2),
        "max_err": m["max_err"], "max_err_non_edge": m["max_err_non_edge"],
        "pixels_changed": float((diff > 0).mean()), "pixels_changed_by_more_than_2": float((diff > 2).mean()),
        "depth_equal": bool(np.array_equal(base["depth"], alt["depth"])),
        "meets_image_tolerance": bool(m["id_match"] >= 0.999 and m["psnr"] >= 45.0 and m["max_err_non_edge"] <= 2),
    }
    if fog is not None:
        out["changed_pixels_inside_fog"] = float((diff[fog] > 0).mean()) if fog.any() else 0.0
        out["changed_pixels_outside_fog"] = float((diff[~fog] > 0).mean()) if (~fog).any() else 0.0
    return out


def standin_medium(w=480, h=270):
    """A frame with texels near the size of a pixel (256^2 textures), where the choice between the
    sampler's two filters (R12, R13) actually varies over the picture."""
    import wgpu_renderer_amd  # noqa: F401
    from wgpu_renderer_amd import abi, scene
    sc = scene.sponza_standin(detail=0.2, tex_size=256)
    rc = scene.REFERENCE_CAMERA
    cam = abi.camera_uniform(rc["position"], rc["pitch"], rc["yaw"], rc["fovy"], w, h)
    return sc, cam, abi.light_at_time(2.0), abi.global_uniforms(2.0, 1), abi.PASS_ALL, w, h, 512


def noise_distance(base, alt, n=4096):
    """fog_noise (fog.wgsl:48-62) at n points of the fog volume under both builds: the share of
    points whose value differs in any bit, and the largest difference."""
    from oracle_lib import f3
    rng = np.random.default_rng(0x5EED)
    pts = rng.uniform([-1360.0, -90.0, -600.0], [1360.0, 60.0, 600.0], (n, 3)).astype(np.float32)
    a = np.array([base.wro_fog_noise(f3(p), 2.0) for p in pts], np.float32)
    b = np.array([alt.wro_fog_noise(f3(p), 2.0) for p in pts], np.float32)
    return {"points": n, "differ_in_bits": float((a.view(np.uint32) != b.view(np.uint32)).mean()), "max_abs_difference": float(np.abs(a - b).max())}


def measure():
    import oracle_lib
    from golden_inputs import cube_config1, standin_small
    frames = {"standin_frame_192x108": standin_small, "standin_frame_480x270_tex256": standin_medium,
              "cube_frame_160x90": lambda: cube_config1(160, 90)}
    with ThreadPoolExecutor(4) as ex:
        paths = dict(zip(ALTERNATIVES, ex.map(build_variant, ALTERNATIVES)))
    result = {"frames": list(frames), "alternatives": {}}
    base = {f: render(None, mk) for f, mk in frames.items()}
    fog = {f: fog_mask(mk) for f, mk in frames.items()}
    result["fog_pixels"] = {f: (None if m is None else float(m.mean())) for f, m in fog.items()}
    for name, (macro, rule, what) in ALTERNATIVES.items():
        L = oracle_lib.typed(paths[name])
        result["alternatives"][name] = {"rule": rule, "macro": macro, "alternative": what,
                                        "fog_noise": noise_distance(oracle_lib.lib(), L),
                                        "distance": {f: distance(base[f], render(L, mk), fog[f]) for f, mk in frames.items()}}
    return result


def main():
    res = measure()
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    for name, a in res["alternatives"].items():
        print(f"{name:22s} fog_noise: {100 * a['fog_noise']['differ_in_bits']:.1f} % of points differ, by at most {a['fog_noise']['max_abs_difference']:.3g}")
        for f, d in a["distance"].items():
            print(f"{name:22s} {f:24s} psnr {d['psnr_db']} dB  max {d['max_err']}  off-edge {d['max_err_non_edge']}  changed {100 * d['pixels_changed']:.2f} %  "
                  f"id {d['id_match']:.4f}  within tolerance: {d['meets_image_tolerance']}")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
