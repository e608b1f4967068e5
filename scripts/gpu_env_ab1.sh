#!/bin/bash
# Developer helper (run through gpurun): the built library with one frame in flight under
# different values of an environment variable ("-" = unset), interleaved, three rounds of 600
# frames.  Usage: bash scripts/gpu_env_ab1.sh VAR v1 v2 ...
VAR=$1; shift
mkdir -p gpurun_out
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 600 --warmup 50 --frames-in-flight ${NF:-1} > gpurun_out/env1_$v.json 2> gpurun_out/env1_$v.err || { echo "FAILED $v"; tail -5 gpurun_out/env1_$v.err; exit 1; }
    python - "$VAR" "$v" "$round" <<'PY'
import json,sys
var,v,r=sys.argv[1:4]
for line in open(f"gpurun_out/env1_{v}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"round {r} {var}={v:>6} fps={d['value']:.1f}", flush=True)
PY
  done
done
