#!/bin/bash
# Profiles of the current build for profiles/ (run on the GPU box through gpurun):
#   bash scripts/gpu_profile_round.sh r03
# writes under gpurun_out/: TAG_kernel_stats.txt (default bench command, kernel trace),
# TAG_one_frame_kernel_stats.txt + TAG_one_frame_timeline.txt (one frame in flight),
# TAG_pmc_sq_{1,4}.txt, TAG_pmc_fetch_{1,4}.txt, TAG_pmc_write_{1,4}.txt, TAG_pmc_ta_tcp_{1,4}.txt (counter passes with
# one and with four frames in flight: the two fog schedules run different kernels) and
# TAG_counters.json (scripts/make_profile_json.py), which bench.py reads.
# Every rocprofv3 run has the program directly after "--" and its own time limit; counter
# passes hold one block's counters each.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
trace() {  # name, frames in flight
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_$1 -- python3 $R/bench.py --steps 100 --warmup 5 --frames-in-flight $2 --no-cpu-baseline --skip-march-alone --no-scale-workloads > $O/${TAG}_$1.log 2>&1 || return 1
  DB=$(find $O/${TAG}_$1 -name "*.db" | head -1)
  python3 $R/scripts/rocpd_summary.py $DB > $O/${TAG}_$1_stats.txt
  python3 $R/scripts/frame_timeline.py $DB > $O/${TAG}_$1_timeline.txt
  rm -rf $O/${TAG}_$1
}
pmc() {  # name, frames in flight, counters
  timeout -s KILL 240 rocprofv3 --pmc $3 -d $O/${TAG}_$1 --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --frames-in-flight $2 --no-cpu-baseline --skip-march-alone --no-scale-workloads > $O/${TAG}_$1.log 2>&1 || return 1
  CSV=$(find $O/${TAG}_$1 -name "*counter_collection.csv" | head -1)
  python3 $R/scripts/pmc_summary.py $CSV > $O/${TAG}_$1.txt
  rm -rf $O/${TAG}_$1
}
trace kernel 4 && trace one_frame 1 || exit 1
for n in 1 4; do
  pmc pmc_sq_$n $n "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" || exit 1
  pmc pmc_fetch_$n $n "FETCH_SIZE" || exit 1
  pmc pmc_write_$n $n "WRITE_SIZE" || exit 1
  pmc pmc_ta_tcp_$n $n "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" || exit 1
done
cd $R && python3 scripts/make_profile_json.py $TAG
