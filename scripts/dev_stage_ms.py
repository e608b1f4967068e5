"""Developer helper: prints per-stage milliseconds of the bench frame for one or more builds
of the library (WR_HIP_LIBRARY is read at import, so each build runs in a child process)."""
import json, os, subprocess, sys
libs = sys.argv[1:] or [""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["WR_HIP_LIBRARY"] = os.path.abspath(lib)
    p = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "3", "--no-cpu-baseline", "--tex-size", "256"],
                       env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(lib, "FAILED", p.stderr[-2000:]); continue
    j = json.loads(line[-1])
    print(lib or "default", "fps=%.1f" % j["value"], {k: round(v, 3) for k, v in j["stage_ms"].items()}, "replay", j["ordered_blend_pixels"], j["ordered_blend_slow_pixels"], j["bin_items"], flush=True)
