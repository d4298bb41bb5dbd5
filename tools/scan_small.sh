#!/bin/bash
# developer loop: small-form GEMM wave shapes, real bench in the default (lock-step) mode and one fit at a time
cd cosmogp_amd/csrc
for v in "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=4" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=2" "-DGEMM_SMALL_WM=1 -DGEMM_SMALL_WN=4" "-DGEMM_SMALL_WM=4 -DGEMM_SMALL_WN=1"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || exit 1
  for w in c3 c2; do
    echo -n "[$v] $w: "
    (cd ../.. && python bench.py --workload $w --steps 32 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['roofline']['by_form']; print(d['value'], d['roofline']['achieved'], [(v['launches'], v['achieved']) for v in f.values()], d['extra']['single_fit_latency_ms'])")
  done
done
