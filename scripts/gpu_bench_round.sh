#!/bin/bash
# Bench outputs of the current build for profiles/ (run on the GPU box through gpurun, after
# scripts/gpu_profile_round.sh has left gpurun_out/TAG_counters.json for this build):
#   bash scripts/gpu_bench_round.sh r02
set -o pipefail
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
[ -f gpurun_out/${TAG}_counters.json ] && cp gpurun_out/${TAG}_counters.json profiles/${TAG}_counters.json
for combo in "WR_DEBUG_SCHEDULE=split" "WR_DEBUG_SCHEDULE=mono" "WR_DEBUG_ONE_STREAM=1" "WR_DEBUG_ONE_STREAM=2"; do
  env $combo timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/dbg_$combo.log 2>&1 || { echo "FAILED $combo"; tail -20 gpurun_out/dbg_$combo.log; exit 1; }
  echo "$combo: $(tail -1 gpurun_out/dbg_$combo.log)"
done
b() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err || { echo "bench $name failed"; tail -5 gpurun_out/${TAG}_bench_$name.err; exit 1; }; echo "bench $name done"; }
b default && b one_frame --frames-in-flight 1 && b two --frames-in-flight 2 && b three --frames-in-flight 3 && b steps20 --steps 20 --warmup 5 && b posebatch --workload posebatch && b tiles4k --workload tiles4k
python scripts/show_bench.py gpurun_out/${TAG}_bench_*.json
