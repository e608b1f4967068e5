#!/bin/bash
# End-of-round GPU run (through gpurun): the parity suite under every debug schedule, then the
# profile round of scripts/gpu_profile_round.sh.  The bench lines kept under profiles/ come from
# a call of their own on a fresh box (scripts/gpu_bench_round.sh): counter passes leave a box slower.
set -o pipefail
cd $GRAFT_REPO_ROOT
for combo in "WR_DEBUG_SCHEDULE=split" "WR_DEBUG_SCHEDULE=mono" "WR_DEBUG_ONE_STREAM=1" "WR_DEBUG_ONE_STREAM=2"; do
  env $combo timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/dbg_$combo.log 2>&1 || { echo "FAILED $combo"; tail -20 gpurun_out/dbg_$combo.log; exit 1; }
  echo "$combo: $(tail -1 gpurun_out/dbg_$combo.log)"
done
bash scripts/gpu_profile_round.sh r03 || { echo "profile round failed"; exit 1; }
