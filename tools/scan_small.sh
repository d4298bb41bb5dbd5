#!/bin/bash
# developer loop: small-form GEMM wave shapes, real bench (single + 4 in flight)
cd cosmogp_amd/csrc
for v in "-DGEMM_SMALL_WM=1 -DGEMM_SMALL_WN=4" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=4" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=2"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || exit 1
  for w in c3 c2; do
    echo -n "[$v] $w: "
    (cd ../.. && python bench.py --workload $w --steps 24 --no-cpu-baseline 2>&1 | grep -oE "\"value\": [0-9.]+|\"single_fit_latency_ms\": [0-9.]+|\"single_fit_cholesky_gflops\": [0-9.]+|\"gemm_update\": [0-9.]+" | tr "\n" " "); echo
  done
done
