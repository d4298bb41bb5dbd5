# configs 2, 3, 5 on the work-bound list: panels a lead-zone tile may lag before it gets a task (CGP_DAG_BLEADLAG), lead columns
for wl in c2 c3 c5; do
for cfg in "0 1" "1 1" "2 1" "3 1" "5 1" "2 2" "4 2" "3 3"; do
  set -- $cfg
  r=$(CGP_DAG_BLEADLAG=$1 CGP_DAG_BLEAD=$2 python3 bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")
  echo "$wl bleadlag=$1 blead=$2 -> $r"
done; done
