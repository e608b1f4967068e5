"""tests/golden/rule_sensitivity.json is what scripts/rule_sensitivity.py measures: the distance
between the oracle's picture and the picture under the other plausible reading of each rule
that nothing else in this repository checks (R1 inside rand, R3, R4, R12, R13's tie, R15's
rounding).  CPU only; the default build of the oracle is untouched by the alternatives."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_recorded_sensitivities_are_what_the_alternative_builds_give():
    import rule_sensitivity
    want = json.load(open(rule_sensitivity.OUT))
    got = json.loads(json.dumps(rule_sensitivity.measure()))
    assert got == want
    alts = want["alternatives"]
    assert set(alts) == set(rule_sensitivity.ALTERNATIVES)
    fogged = "standin_frame_192x108"
    # every alternative was really compiled in: each one changes something it should change
    assert alts["R1_fused_dot"]["fog_noise"]["differ_in_bits"] > 0.5
    assert alts["R4_unfused_sine"]["fog_noise"]["differ_in_bits"] > 0.5
    assert 0.0 < alts["R3_mix_one_mul"]["fog_noise"]["max_abs_difference"] < 1e-6
    assert alts["R15_truncate"]["distance"][fogged]["pixels_changed"] > 0.5
    # the hash amplifies R1 and R4 into other noise, and only where there is fog
    for name in ("R1_fused_dot", "R4_unfused_sine"):
        d = alts[name]["distance"][fogged]
        assert d["changed_pixels_outside_fog"] == 0.0 and d["changed_pixels_inside_fog"] > 0.0 and d["depth_equal"]
        assert alts[name]["distance"]["cube_frame_160x90"]["pixels_changed"] == 0.0  # no fog pass in that frame
    # geometry (ids, depth) depends on none of these rules
    for a in alts.values():
        for d in a["distance"].values():
            assert d["id_match"] == 1.0 and d["depth_equal"]
