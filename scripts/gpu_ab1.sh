#!/bin/bash
# Developer helper (run through gpurun): one frame in flight only, library variants from
# build_dev/ interleaved, three rounds of 600 frames each.  Usage: bash scripts/gpu_ab1.sh name1 name2 ...
mkdir -p gpurun_out
for round in 1 2 3; do
  for v in "$@"; do
    WR_HIP_LIBRARY=$PWD/build_dev/libwr_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 600 --warmup 50 --frames-in-flight ${NF:-1} > gpurun_out/ab1_${v}.json 2> gpurun_out/ab1_${v}.err || { echo "FAILED $v"; tail -5 gpurun_out/ab1_${v}.err; exit 1; }
    python - "$v" "$round" <<'PY'
import json,sys
v,r=sys.argv[1:3]
for line in open(f"gpurun_out/ab1_{v}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"round {r} {v:>10} fps={d['value']:.1f}", flush=True)
PY
  done
done
