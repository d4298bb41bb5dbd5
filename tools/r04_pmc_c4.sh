#!/bin/bash
# ./tools/r04_pmc_c4.sh <tag> [extra bench.py arguments]: HBM traffic per pass of config 4 (bench.py's default objects per pass, 512 x N = 512, M = 19) on ONE handle,
# two rocprofv3 --pmc passes -> gpurun_out/<tag>.json (tools/pmc_pass_traffic.py).  Run on the GPU box from the repo root.
tag=${1:-r04_pmc_traffic_c4}; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out
args="--workload c4 --streams 1 --no-extras --no-cpu-baseline --steps 20 --warmup 0 $*"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o run --output-format csv -- python3 $root/bench.py $args > /tmp/pmc_$c.log 2>&1) || { tail -5 /tmp/pmc_$c.log; exit 1; }
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_pass_traffic.py $f $w ${CGP_BENCH_C4_NOBJ:-512} gpurun_out/$tag.json
