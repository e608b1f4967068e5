#!/bin/bash
# Developer helper (GPU box): library variants of build_dev/ under the default bench command's
# timing (100 frames after 5) and under the driver's (20 after 5), four frames in flight, two rounds.
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    for steps in 100 20; do
      WR_HIP_LIBRARY=$PWD/build_dev/libwr_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps $steps --warmup 5 > gpurun_out/abd_${v}.json 2> gpurun_out/abd_${v}.err || { echo "FAILED $v"; tail -5 gpurun_out/abd_${v}.err; exit 1; }
      python - "$v" "$round" "$steps" <<'PY'
import json,sys
v,r,st=sys.argv[1:4]
d=json.loads(open(f"gpurun_out/abd_{v}.json").read().strip().splitlines()[-1])
print(f"round {r} {v:>10} steps={st} fps={d['value']:.1f}", flush=True)
PY
    done
  done
done
