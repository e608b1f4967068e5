#!/bin/bash
# Developer helper (run through gpurun): benches library variants from build_dev/ back to back on
# one box, four frames and one frame in flight.  Usage: bash scripts/gpu_ab.sh name1 name2 ...
mkdir -p gpurun_out
for v in "$@"; do
  for nf in 4 1; do
    WR_HIP_LIBRARY=$PWD/build_dev/libwr_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --skip-march-alone --no-scale-workloads --steps 200 --warmup 20 --frames-in-flight $nf > gpurun_out/ab_${v}_$nf.json 2> gpurun_out/ab_${v}_$nf.err || { echo "FAILED $v $nf"; tail -5 gpurun_out/ab_${v}_$nf.err; exit 1; }
    python - "$v" "$nf" <<'PY'
import json,sys
v,nf=sys.argv[1:3]
for line in open(f"gpurun_out/ab_{v}_{nf}.json"):
    if line.startswith("{"):
        d=json.loads(line); print(f"{v:>10} in_flight={nf} fps={d['value']:.1f} march_ms={d['roofline'].get('launch_ms')}", flush=True)
PY
  done
done
