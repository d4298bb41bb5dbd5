#!/bin/bash
# ./tools/final_pass_r04.sh <part 1|2|3> : the measurements behind profiles/r04_* (run on the GPU box from the repository root)
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
if [ "$1" = "1" ]; then
  # the driver's command (default line: config 3 + extra.configs = c2 / c4 / c5 by child processes) and the four workloads on their own
  timeout -k 10 600 python3 bench.py > $out/r04_bench_final_c3.json 2> $out/bench_c3.err || echo "bench c3 failed"
  echo "bench c3 (default line) done"
  for w in c2 c4 c5; do
    timeout -k 10 400 python3 bench.py --workload $w > $out/r04_bench_final_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
    echo "bench $w done"
  done
  timeout -k 10 200 python3 bench.py --batch 1 --streams 1 --no-cpu-baseline > $out/r04_bench_c3_single.json 2>> $out/bench_single.err
  echo "bench single done"
elif [ "$1" = "2" ]; then
  # rocprofv3 kernel statistics of the default command, of one fit at a time and of config 4 (no extras: only the timed configuration runs)
  (cd /tmp && rm -rf /tmp/ks_def && rocprofv3 --kernel-trace --stats -d /tmp/ks_def -o run --output-format csv -- python3 $root/bench.py --steps 3 --warmup 0 --no-extras --no-cpu-baseline > $root/$out/r04_bench_c3_noextras.json 2> $root/$out/ks_def.err)
  cp $(find /tmp/ks_def -name "*kernel_stats.csv" | head -1) $out/r04_rocprofv3_kernel_stats_c3_noextras.csv
  echo "stats default done"
  (cd /tmp && rm -rf /tmp/ks_one && rocprofv3 --kernel-trace --stats -d /tmp/ks_one -o run --output-format csv -- python3 $root/bench.py --batch 1 --streams 1 --no-extras --no-cpu-baseline > $root/$out/r04_bench_c3_single_noextras.json 2> $root/$out/ks_one.err)
  cp $(find /tmp/ks_one -name "*kernel_stats.csv" | head -1) $out/r04_rocprofv3_kernel_stats_c3_single.csv
  echo "stats single done"
  (cd /tmp && rm -rf /tmp/ks_c4 && rocprofv3 --kernel-trace --stats -d /tmp/ks_c4 -o run --output-format csv -- python3 $root/bench.py --workload c4 --steps 20 --warmup 0 --no-extras --no-cpu-baseline > $root/$out/r04_bench_c4_noextras.json 2> $root/$out/ks_c4.err)
  cp $(find /tmp/ks_c4 -name "*kernel_stats.csv" | head -1) $out/r04_rocprofv3_kernel_stats_c4_noextras.csv
  echo "stats c4 done"
else
  # HBM traffic (PMC, separate passes): default command (k_chol_dag, 8 objects per launch), configs 2 / 5, config 4 per pass
  args="--no-extras --no-cpu-baseline --steps 3 --warmup 0"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o run --output-format csv -- python3 $root/bench.py $args > /tmp/pmc_$c.log 2>&1) || { tail -5 /tmp/pmc_$c.log; exit 1; }
  done
  python3 tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 8 $out/r04_pmc_traffic_c3_dagbatch.json k_chol_dag
  echo "traffic c3 done"
  cp tools/pmc_run.sh /tmp/pmc_run_r04.sh
  ./tools/pmc_run.sh final/r04_pmc_traffic_c2_dagbatch 16 k_chol_dag --workload c2 && echo "traffic c2 done"
  ./tools/pmc_run.sh final/r04_pmc_traffic_c5_dagbatch 4 k_chol_dag --workload c5 && echo "traffic c5 done"
  bash tools/r04_pmc_c4.sh final/r04_pmc_traffic_c4 && echo "traffic c4 done"
  ./tools/pmc_run.sh final/r04_pmc_traffic_c3_single 1 k_chol_dag --batch 1 --streams 1 && echo "traffic c3 single done"
fi
