# configs 2 and 3 on the work-bound list (window 1, lead 1, depth 16 / 24): the remaining knobs -- fill, longest far update, order
for wl in c2 c3; do
for cfg in "70 0 1" "40 0 1" "100 0 1" "150 0 1" "70 32 1" "70 48 1" "70 0 0" "25 0 1"; do
  set -- $cfg
  r=$(CGP_DAG_BFILL=$1 CGP_DAG_BDMAX=$2 CGP_DAG_BORDER=$3 python3 bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")
  echo "$wl bfill=$1 bdmax=$2 border=$3 -> $r"
done; done
