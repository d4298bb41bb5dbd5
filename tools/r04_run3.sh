#!/bin/bash
mkdir -p gpurun_out
echo "== eigen test"; timeout -k 10 600 python -m pytest tests/test_eigen.py -m gpu -x -q -k "effective" 2>&1 | tail -3
echo "== c4 line"; python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r04_bench_c4_b.json 2> gpurun_out/r04_bench_c4_b.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r04_bench_c4_b.json')); print(d['value'], d['roofline'].get('frac_end_to_end'), {k:v for k,v in d['extra'].items() if ('api' in k or 'small' in k or '1024' in k) and k != 'api_note'})"
echo "== c4 line, no packing"; CGP_NO_PACK=1 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['extra']['small_objects'])"
echo "== default line"; python bench.py > gpurun_out/r04_bench_c3_a.json 2> gpurun_out/r04_bench_c3_a.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r04_bench_c3_a.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_end_to_end'), d['roofline'].get('traffic_source_current')); print(json.dumps(d['extra'].get('configs'), indent=1)[:3000])"
