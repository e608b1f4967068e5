#!/bin/bash
# Developer helper (run through gpurun): both fog schedules at 1..4 frames in flight.
mkdir -p gpurun_out
for nf in 4 3 2 1; do
  for sc in mono split; do
    WR_DEBUG_SCHEDULE=$sc timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 200 --warmup 20 --frames-in-flight $nf > gpurun_out/sched_${sc}_$nf.json 2> gpurun_out/sched_${sc}_$nf.err || { echo "FAILED $sc $nf"; tail -5 gpurun_out/sched_${sc}_$nf.err; exit 1; }
    python - $sc $nf <<'PY'
import json,sys
sc,nf=sys.argv[1:3]
for line in open(f"gpurun_out/sched_{sc}_{nf}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"{sc:>6} in_flight={nf} fps={d['value']:.1f}", flush=True)
PY
  done
done
