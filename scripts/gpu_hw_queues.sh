#!/bin/bash
# Frame rate against the number of HIP hardware queues per process (GPU_MAX_HW_QUEUES) and frames in flight.
set -e
out=gpurun_out/r3_hw_queues.txt
: > $out
for q in 4 8 16; do
  for n in 1 2 4; do
    timeout -k 10 120 python scripts/dev_ramp.py $n 0 $q 2>/dev/null | grep -E "300 frames|complete" >> $out
  done
done
