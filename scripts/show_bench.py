"""Prints the headline fields of a bench.py output file."""
import json
import sys
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            r = d.get("roofline") or {}
            print(path, round(d["value"], 1), d["unit"], "stage_ms", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get("stage_ms", {}).items() if k != "note"},
                  "launch_ms", r.get("launch_ms"), "frac", r.get("frac"), "one_frame", (r.get("one_frame_in_flight") or {}).get("frames_per_s"))
