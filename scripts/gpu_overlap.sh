#!/bin/bash
# Kernel trace of bench.py with N frames in flight: overlap statistics (scripts/dev_timeline.py).
TAG=$1; N=${2:-4}; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/$TAG -- python3 $R/bench.py --steps 60 --warmup 5 --frames-in-flight $N --no-cpu-baseline --skip-march-alone --no-scale-workloads "$@" > $R/gpurun_out/$TAG.log 2>&1 || exit 1
cd $R
DB=$(find gpurun_out/$TAG -name "*.db" | head -1)
python scripts/dev_timeline.py $DB "k_shade<false" > gpurun_out/${TAG}_overlap.txt
python scripts/rocpd_summary.py $DB > gpurun_out/${TAG}_stats.txt
grep -o '"value": [0-9.]*' gpurun_out/$TAG.log
rm -rf gpurun_out/$TAG
cat gpurun_out/${TAG}_overlap.txt
