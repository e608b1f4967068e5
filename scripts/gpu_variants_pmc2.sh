#!/bin/bash
# Developer helper (GPU box): like gpu_variants_pmc.sh with a counter list of the caller's.
# Usage: bash scripts/gpu_variants_pmc2.sh PATTERN "COUNTERS" name1 name2 ...
PAT=$1; CTRS=$2; shift; shift
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so
  bash $R/scripts/gpu_pmc.sh dev_pmc2_$v "$CTRS" > /dev/null 2>&1 || { echo "$v failed"; exit 1; }
  echo "== $v"; head -1 $R/gpurun_out/dev_pmc2_$v.txt | cut -c1-200; grep -E "$PAT" $R/gpurun_out/dev_pmc2_$v.txt | grep -v "true>" | cut -c1-200
done
