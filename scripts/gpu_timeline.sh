#!/bin/bash
# Kernel trace of `bench.py --frames-in-flight N` on the GPU box: per-kernel statistics and the
# timeline of one steady-state frame.  Usage (through gpurun): bash scripts/gpu_timeline.sh TAG [N] [extra bench args]
TAG=$1; N=${2:-1}; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -- python3 $R/bench.py --steps 40 --warmup 5 --frames-in-flight $N --no-cpu-baseline --skip-march-alone --no-scale-workloads "$@" > $R/gpurun_out/$TAG.log 2>&1 || exit 1
cd $R
DB=$(find gpurun_out/$TAG -name "*.db" | head -1)
python scripts/frame_timeline.py $DB > gpurun_out/${TAG}_timeline.txt
python scripts/rocpd_summary.py $DB > gpurun_out/${TAG}_stats.txt
grep -o '"value": [0-9.]*' gpurun_out/$TAG.log
rm -rf gpurun_out/$TAG
