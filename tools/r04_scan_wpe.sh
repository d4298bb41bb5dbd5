#!/bin/bash
# config 4 (512 per pass) on one handle + 2 handles for builds of the 64 x 64 update form at a minimum of 1 (default) / 6 / 8 waves
# per SIMD (rebuilds the library on the box; the last build is the default again)
cd cosmogp_amd/csrc
for v in "" "-DCGP_SMALL_WPE=6" "-DCGP_SMALL_WPE=8" ""; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-unused-function $v -o ../libcosmogp_hip.so api.hip || { echo "build failed: $v"; continue; }
  echo "[$v]"
  (cd ../.. && python tools/c4_ab.py 512 && python3 bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   2 handles:', d['value'], d['ms_per_step'])")
done
