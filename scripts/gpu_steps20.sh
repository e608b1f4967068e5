#!/bin/bash
# Developer helper (GPU box): the driver's command (20 timed frames after 5) for every number of
# frames in flight, three rounds each, as separate processes like the driver's.
mkdir -p gpurun_out
for round in 1 2 3; do
  for nf in 1 2 3 4; do
    timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-march-alone --no-scale-workloads --frames-in-flight $nf > gpurun_out/s20_$nf.json 2> gpurun_out/s20_$nf.err || { echo "FAILED $nf"; tail -3 gpurun_out/s20_$nf.err; exit 1; }
    python - "$nf" "$round" <<'PY'
import json,sys
nf,r=sys.argv[1:3]
d=json.loads(open(f"gpurun_out/s20_{nf}.json").read().strip().splitlines()[-1])
print(f"round {r} in_flight={nf} fps={d['value']:.1f} ms_per_step={d['ms_per_step']:.4f}", flush=True)
PY
  done
done
