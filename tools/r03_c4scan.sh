# config 4: objects per launch sequence x handles in flight
for cfg in "128 1" "128 2" "128 4" "128 6" "64 4" "64 8" "256 2" "256 4" "512 2" "96 4" "192 4"; do
  set -- $cfg
  r=$(CGP_BENCH_C4_NOBJ=$1 python3 bench.py --workload c4 --streams $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "B=$1 handles=$2 -> $r"
done
