#!/bin/bash
# Developer helper (run through gpurun): kernel trace and SQ counters of the fog noise kernels for
# library variants from build_dev/, one frame in flight.  Usage: bash scripts/gpu_variant_prof.sh name1 name2 ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export WR_HIP_LIBRARY=$R/build_dev/libwr_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/vp_$v -- python3 $R/bench.py --steps 50 --warmup 5 --frames-in-flight 1 --no-cpu-baseline --skip-march-alone --no-scale-workloads > $O/vp_$v.log 2>&1 || exit 1
  DB=$(find $O/vp_$v -name "*.db" | head -1)
  python3 $R/scripts/rocpd_summary.py $DB > $O/vp_${v}_stats.txt
  rm -rf $O/vp_$v
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $O/vq_$v --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --frames-in-flight 1 --no-cpu-baseline --skip-march-alone --no-scale-workloads > $O/vq_$v.log 2>&1 || exit 1
  CSV=$(find $O/vq_$v -name "*counter_collection.csv" | head -1)
  python3 $R/scripts/pmc_summary.py $CSV > $O/vq_$v.txt
  rm -rf $O/vq_$v
  echo "== $v"; grep -E "^kernel|k_fog_noise" $O/vp_${v}_stats.txt; grep -E "^kernel|k_fog_noise.*false" $O/vq_$v.txt
done
