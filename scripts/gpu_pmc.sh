#!/bin/bash
# One rocprofv3 counter pass over `bench.py --frames-in-flight 1` on the GPU box (the profiler
# serialises the kernels): mean counter values per kernel.
# Usage (through gpurun): bash scripts/gpu_pmc.sh TAG "COUNTER COUNTER ..." [extra bench args]
TAG=$1; CTRS=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS -d $R/gpurun_out/$TAG --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --frames-in-flight 1 --no-cpu-baseline --skip-march-alone --no-scale-workloads "$@" > $R/gpurun_out/$TAG.log 2>&1 || { tail -5 $R/gpurun_out/$TAG.log; exit 1; }
cd $R
CSV=$(find gpurun_out/$TAG -name "*counter_collection.csv" | head -1)
python scripts/pmc_summary.py $CSV > gpurun_out/${TAG}.txt
rm -rf gpurun_out/$TAG
head -14 gpurun_out/${TAG}.txt | cut -c1-250
