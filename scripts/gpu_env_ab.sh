#!/bin/bash
# Developer helper (run through gpurun): benches the built library with one frame in flight under
# different values of an environment variable.  Usage: bash scripts/gpu_env_ab.sh VAR v1 v2 ...
VAR=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $VAR=$v timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 200 --warmup 20 --frames-in-flight 1 > gpurun_out/env_$v.json 2> gpurun_out/env_$v.err || { echo "FAILED $v"; tail -5 gpurun_out/env_$v.err; exit 1; }
  python - "$VAR" "$v" <<'PY'
import json,sys
var,v=sys.argv[1:3]
for line in open(f"gpurun_out/env_{v}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"{var}={v:>4} fps={d['value']:.1f}", flush=True)
PY
done
