#!/bin/bash
# Developer helper (run through gpurun): standalone kernel times (every kernel alone on one stream,
# WR_DEBUG_ONE_STREAM=1) of library variants from build_dev/, for both fog schedules.
# Usage: bash scripts/gpu_variant_alone.sh name1 name2 ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export WR_DEBUG_ONE_STREAM=1
for v in "$@"; do
  export WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so
  for nf in 1 4; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/va_${v}_$nf -- python3 $R/bench.py --steps 30 --warmup 5 --frames-in-flight $nf --no-cpu-baseline --skip-march-alone --no-scale-workloads > $O/va_${v}_$nf.log 2>&1 || exit 1
    DB=$(find $O/va_${v}_$nf -name "*.db" | head -1)
    python3 $R/scripts/rocpd_summary.py $DB > $O/va_${v}_${nf}_stats.txt
    rm -rf $O/va_${v}_$nf
    echo "== $v, $nf in flight"; grep -E "k_fog_noise_(view|light)<false>|k_fog_march<false>|k_shade<false>|k_raster" $O/va_${v}_${nf}_stats.txt
  done
done
