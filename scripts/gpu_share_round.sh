#!/bin/bash
# Developer helper (GPU box): parity of one variant library, then frame rates and VALU counts of several.
# Usage: bash scripts/gpu_share_round.sh name1 name2 ...
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  [ "$v" = share0 ] && continue
  WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 150 --timeout-method thread -m gpu -k "all_passes or golden or camera_poses or frames_in_flight or fog_volume or full_size_frame or light_animation" > gpurun_out/share_parity_$v.log 2>&1 || { echo "PARITY FAILED $v"; tail -20 gpurun_out/share_parity_$v.log; exit 1; }
  echo "parity ok $v"
done
bash scripts/gpu_ab.sh "$@" && bash scripts/gpu_variants_pmc.sh "k_fog_noise|k_fog_march" "$@"
