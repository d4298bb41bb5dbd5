#!/bin/bash
# developer loop: rebuild the library with different GEMM wave shapes on the GPU box and bench (4 fits in flight)
cd cosmogp_amd/csrc
for v in "-DGEMM_BIG_WM=4 -DGEMM_BIG_WN=2" "-DGEMM_BIG_WM=2 -DGEMM_BIG_WN=4" "-DGEMM_SMALL_WM=1 -DGEMM_SMALL_WN=4 -DGEMM_KB_SMALL=32" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=4" "-DGEMM_KB_BIG=32" "-DGEMM_BIG_TILE_THRESHOLD=1024 -DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=2"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || exit 1
  for w in c3 c2; do
    echo -n "[$v] $w: "
    (cd ../.. && python bench.py --workload $w --steps 16 --no-cpu-baseline 2>&1 | grep -oE "\"value\": [0-9.]+|\"single_fit_latency_ms\": [0-9.]+" | tr "\n" " "); echo
  done
done
