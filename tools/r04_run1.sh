#!/bin/bash
# round 4, GPU call 1: full GPU suite, config 4 with / without the XCD-affine launches, its PMC traffic
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04_tests1.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04_tests1.log
grep -h "bound =" gpurun_out/r04_tests1.log | head -20
echo "== c4 one handle"; python tools/c4_ab.py 128 CGP_NO_XCD=1 2>&1 | tee gpurun_out/r04_c4_ab.log
python tools/c4_ab.py 256 CGP_NO_XCD=1 2>&1 | tee -a gpurun_out/r04_c4_ab.log
echo "== c4 bench, 4 handles"
for env in "" "CGP_NO_XCD=1"; do
  for st in 1 2 4; do
    r=$(env $env python3 bench.py --workload c4 --streams $st --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "env=[$env] handles=$st -> $r" | tee -a gpurun_out/r04_c4_ab.log
  done
done
echo "== pmc c4"
bash tools/r04_pmc_c4.sh r04_pmc_traffic_c4_xcd
CGP_NO_XCD=1 bash tools/r04_pmc_c4.sh r04_pmc_traffic_c4_plain
