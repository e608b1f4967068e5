#!/bin/bash
# Developer helper (GPU box): for each variant library build_dev/libwr_NAME.so, the serialised
# duration (GRBM_GUI_ACTIVE / 8 XCDs, cycles) and VALU instructions of the kernels matching PATTERN.
# Usage: bash scripts/gpu_variants_pmc.sh PATTERN name1 name2 ...
PAT=$1; shift
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so
  bash $R/scripts/gpu_pmc.sh dev_pmc_$v "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" > /dev/null 2>&1 || { echo "$v failed"; exit 1; }
  echo "== $v"; grep -E "$PAT" $R/gpurun_out/dev_pmc_$v.txt | grep -v "k_fog[a-z_]*<true>\|k_shade<true>\|k_replay_slow<true>" | cut -c1-120
done
