#!/bin/bash
# Developer helper (run through gpurun): frames/s against the number of frames in flight.
mkdir -p gpurun_out
for nf in "$@"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 300 --warmup 30 --frames-in-flight $nf > gpurun_out/nf_$nf.json 2> gpurun_out/nf_$nf.err || { echo "FAILED $nf"; tail -5 gpurun_out/nf_$nf.err; exit 1; }
  python - $nf <<'PY'
import json,sys
nf=sys.argv[1]
for line in open(f"gpurun_out/nf_{nf}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"in_flight={nf} fps={d['value']:.1f} march_ms={d['roofline']['launch_ms']:.3f}", flush=True)
PY
done
