#!/bin/bash
# ./tools/final_pass.sh <part 1|2> : the measurements behind profiles/r02_* (run on the GPU box from the repository root)
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
if [ "$1" = "1" ]; then
  for w in c3 c2 c4 c5; do
    timeout -k 10 280 python3 bench.py --workload $w > $out/r02_bench_final_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
    echo "bench $w done"
  done
  timeout -k 10 200 python3 bench.py --batch 1 --streams 1 --no-cpu-baseline > $out/r02_bench_c3_single.json 2>> $out/bench_single.err
  echo "bench single done"
else
  # rocprofv3 kernel statistics of the default command and of one fit at a time (no extras: only the timed configuration runs)
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks_def -o run --output-format csv -- python3 $root/bench.py --steps 16 --warmup 0 --no-extras --no-cpu-baseline > $root/$out/r02_bench_c3_noextras.json 2> $root/$out/ks_def.err)
  cp $(find /tmp/ks_def -name "*kernel_stats.csv" | head -1) $out/r02_rocprofv3_kernel_stats_c3_noextras.csv
  echo "stats default done"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks_one -o run --output-format csv -- python3 $root/bench.py --batch 1 --streams 1 --no-extras --no-cpu-baseline > $root/$out/r02_bench_c3_single_noextras.json 2> $root/$out/ks_one.err)
  cp $(find /tmp/ks_one -name "*kernel_stats.csv" | head -1) $out/r02_rocprofv3_kernel_stats_c3_single.csv
  echo "stats single done"
  ./tools/pmc_run.sh final/r02_pmc_traffic_c3_dagbatch 8 k_chol_dag --workload c3 && echo "pmc default done"
  ./tools/pmc_run.sh final/r02_pmc_traffic_c3_single 1 k_chol_dag --workload c3 --batch 1 --streams 1 && echo "pmc single done"
fi
