#!/bin/bash
# ./tools/pmc_run.sh <out name> <objects per launch> <kernel substring> <bench.py arguments ...>
# two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench.py command -> gpurun_out/<out name>.json (tools/pmc_traffic.py)
name=$1; objs=$2; ksub=$3; shift 3
root=$(pwd)
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o run --output-format csv -- python3 $root/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1) || { tail -5 /tmp/pmc_$c.log; exit 1; }
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $f $w $objs gpurun_out/$name.json $ksub
