#!/bin/bash
# config 4 (512 per pass) on one handle + 2 handles for several builds of the small update form (rebuilds the library on the box)
cd cosmogp_amd/csrc
for v in "" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=2 -DCGP_EARLYC_MAX=16" "-DGEMM_SMALL_WM=1 -DGEMM_SMALL_WN=4 -DCGP_EARLYC_MAX=16" "-DGEMM_SMALL_WM=2 -DGEMM_SMALL_WN=2 -DCGP_EARLYC_MAX=8"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || { echo "build failed: $v"; continue; }
  echo "[$v]"
  (cd ../.. && python tools/c4_ab.py 512 && python3 bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   2 handles:', d['value'], d['ms_per_step'])")
done
