#!/bin/bash
# round 4, GPU call 2: new tests (XCD, pipelined, packed, integration doc, eigen threshold), c4 line with the new extras, default line with extra.configs
mkdir -p gpurun_out
echo "== new tests"; timeout -k 10 900 python -m pytest tests/test_round2_gpu.py tests/test_integration_doc.py tests/test_eigen.py -m gpu -x -q -k "xcd or pipelined or packed or integration or effective or first_touch" > gpurun_out/r04_tests2.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r04_tests2.log
echo "== c4 line"; python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r04_bench_c4_a.json 2> gpurun_out/r04_bench_c4_a.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r04_bench_c4_a.json')); print(d['value'], d['roofline'].get('frac_end_to_end'), {k:v for k,v in d['extra'].items() if 'api' in k or 'small' in k or '1024' in k})"
