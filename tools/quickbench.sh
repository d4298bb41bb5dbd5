#!/bin/bash
# quick GPU check: parity tests + stage timings for c3 and c2 (developer loop)
set -o pipefail
timeout -k 10 400 python -m pytest tests -q -x -m gpu > gpurun_out/qb_tests.log 2>&1; echo "tests exit=$?"; tail -2 gpurun_out/qb_tests.log
for w in c3 c2; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline > gpurun_out/qb_$w.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/qb_$w.log') if x.startswith('{')]
if not l: print(open('gpurun_out/qb_$w.log').read()[-2000:])
else:
    d=json.loads(l[-1]); print('$w', 'ms/step', d['ms_per_step'], 'gemm TF', d['roofline']['achieved'], d['extra']['stage_ms'])
PY
done
