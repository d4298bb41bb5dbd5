#!/bin/bash
# end-of-round pass on the GPU box: the GPU suite, the remaining traffic constants (config 4 per pass, one fit at a time), the stress tools
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_suite.log 2>&1 || { tail -20 gpurun_out/final/gpu_suite.log; exit 1; }
tail -2 gpurun_out/final/gpu_suite.log
bash tools/r04_pmc_c4.sh final/r04_pmc_traffic_c4 && echo "traffic c4 done"
./tools/pmc_run.sh final/r04_pmc_traffic_c3_single 1 k_chol_dag --batch 1 --streams 1 && echo "traffic c3 single done"
L=gpurun_out/final/r04_stress_end_of_round.log
{ echo "== fused_stress 60 s"; timeout -k 10 150 python tools/fused_stress.py 60 | tail -1
  echo "== dag_stress 120 s"; timeout -k 10 240 python tools/dag_stress.py 120 | tail -1
  echo "== dag_concurrent_stress 60 s x 4 threads"; timeout -k 10 200 python tools/dag_concurrent_stress.py 60 4 | tail -1; } > $L 2>&1
cat $L
