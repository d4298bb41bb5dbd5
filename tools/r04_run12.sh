#!/bin/bash
echo "== c4 one handle"; python tools/c4_ab.py 128 CGP_NO_TALLSEP=1
echo "== ll_check (opt-in LL vs default)"; python tools/ll_check.py 2>&1 | grep -E "bit-identical|worst|oracle" | head -12
for env in "" "CGP_NO_TALLSEP=1"; do
  for st in 1 4; do
    r=$(env $env python3 bench.py --workload c4 --streams $st --steps 400 --warmup 40 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "env=[$env] handles=$st -> $r"
  done
done
