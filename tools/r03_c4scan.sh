for cfg in "128 4" "128 2" "64 4" "64 8" "32 8" "32 16" "16 16" "256 2" "256 4"; do
  set -- $cfg
  r=$(CGP_BENCH_C4_NOBJ=$1 python3 bench.py --workload c4 --streams $2 --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "B=$1 streams=$2 -> $r"
done
