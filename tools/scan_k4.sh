#!/bin/bash
# developer loop: rebuild with different trsm workgroup sizes on the GPU box, parity-check, bench
cd cosmogp_amd/csrc
for v in "-DCGP_K4_WAVES=4" "-DCGP_K4_WAVES=8"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || exit 1
  (cd ../.. && timeout -k 10 300 python -m pytest tests -q -x -m gpu -k "golden or c2 or batch" 2>&1 | tail -1)
  for w in c3 c2 c4; do
    echo -n "[$v] $w: "
    (cd ../.. && python bench.py --workload $w --steps 16 --no-cpu-baseline 2>&1 | grep -oE "\"value\": [0-9.]+|\"single_fit_latency_ms\": [0-9.]+|\"stage_ms\": \{[^}]*\}" | tr "\n" " "); echo
  done
done
