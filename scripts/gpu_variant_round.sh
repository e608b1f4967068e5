#!/bin/bash
# Developer helper (GPU box): parity of each variant library of build_dev/ but the first, then
# frame rates (four and one frames in flight) and the noise kernels' cycle / VALU counts of all.
# Usage: bash scripts/gpu_variant_round.sh base name1 name2 ...
R=$GRAFT_REPO_ROOT
first=1
for v in "$@"; do
  if [ $first = 1 ]; then first=0; continue; fi
  WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 150 --timeout-method thread -m gpu -k "all_passes or golden or camera_poses or frames_in_flight or fog_volume or full_size_frame or light_animation or ragged" > gpurun_out/variant_parity_$v.log 2>&1 || { echo "PARITY FAILED $v"; tail -20 gpurun_out/variant_parity_$v.log; exit 1; }
  echo "parity ok $v"
done
bash scripts/gpu_ab.sh "$@" && bash scripts/gpu_variants_pmc.sh "${PMC_PATTERN:-k_fog_noise|k_fog_march}" "$@"
