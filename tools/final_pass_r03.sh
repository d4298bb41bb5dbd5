#!/bin/bash
# ./tools/final_pass_r03.sh <part 1|2> : the measurements behind profiles/r03_* (run on the GPU box from the repository root)
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
if [ "$1" = "1" ]; then
  for w in c3 c2 c4 c5; do
    timeout -k 10 400 python3 bench.py --workload $w > $out/r03_bench_final_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
    echo "bench $w done"
  done
  timeout -k 10 200 python3 bench.py --batch 1 --streams 1 --no-cpu-baseline > $out/r03_bench_c3_single.json 2>> $out/bench_single.err
  echo "bench single done"
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > $out/r03_bench_c3_as_rank_world1.json 2> $out/bench_rank.err
  echo "bench as rank done"
else
  # rocprofv3 kernel statistics of the default command and of one fit at a time (no extras: only the timed configuration runs)
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks_def -o run --output-format csv -- python3 $root/bench.py --steps 3 --warmup 0 --no-extras --no-cpu-baseline > $root/$out/r03_bench_c3_noextras.json 2> $root/$out/ks_def.err)
  cp $(find /tmp/ks_def -name "*kernel_stats.csv" | head -1) $out/r03_rocprofv3_kernel_stats_c3_noextras.csv
  echo "stats default done"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks_one -o run --output-format csv -- python3 $root/bench.py --batch 1 --streams 1 --no-extras --no-cpu-baseline > $root/$out/r03_bench_c3_single_noextras.json 2> $root/$out/ks_one.err)
  cp $(find /tmp/ks_one -name "*kernel_stats.csv" | head -1) $out/r03_rocprofv3_kernel_stats_c3_single.csv
  echo "stats single done"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks_c4 -o run --output-format csv -- python3 $root/bench.py --workload c4 --steps 20 --warmup 0 --no-extras --no-cpu-baseline > $root/$out/r03_bench_c4_noextras.json 2> $root/$out/ks_c4.err)
  cp $(find /tmp/ks_c4 -name "*kernel_stats.csv" | head -1) $out/r03_rocprofv3_kernel_stats_c4_noextras.csv
  echo "stats c4 done"
  ./tools/r03_pmc.sh final/r03 && echo "pmc done"
  # config 4 again with a timed region of ~80 ms (the default 10 steps are 4 ms: start-up effects of four host threads show)
  timeout -k 10 400 python3 bench.py --workload c4 > $out/r03_bench_final_c4.json 2> $out/bench_c4b.err && echo "bench c4 (200 steps) done"
fi
