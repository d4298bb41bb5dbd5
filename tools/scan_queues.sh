#!/bin/bash
# developer scan: hardware queues x fits in flight x (second stream per handle or not)
w=${1:-c2}
for s2 in 0 1; do
for q in 2 3 4 6 8; do
for s in 2 3 4 5 6 8; do
  if [ $s2 = 1 ]; then export CGP_STREAM2=1; else unset CGP_STREAM2; fi
  v=$(GPU_MAX_HW_QUEUES=$q python bench.py --workload $w --streams $s --steps 48 --no-cpu-baseline --no-extras 2>&1 | grep -oE "\"value\": [0-9.]+" | cut -d" " -f2)
  echo "stream2=$s2 HWQ=$q fits_in_flight=$s $w: $v"
done; done; done
