#!/bin/bash
mkdir -p gpurun_out
echo "== full gpu suite"; timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r04_tests_final_b.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04_tests_final_b.log
echo "== api profile"; python tools/api_profile.py 2>&1 | tail -8
echo "== c4 line"; python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:v for k,v in d['extra'].items() if 'api' in k and k!='api_note'}, d['extra']['small_objects']['api_fits_per_sec'])"
