#!/bin/bash
mkdir -p gpurun_out
bash tools/final_pass_r04.sh 3 2>&1 | grep -E "done|traffic|fits"
bash tools/final_pass_r04.sh 1
echo "== stress"
{
echo "== fused_stress 60 s"; timeout -k 10 200 python tools/fused_stress.py 60 2>&1 | tail -3
echo "== dag_stress 120 s"; timeout -k 10 300 python tools/dag_stress.py 120 41 2>&1 | tail -3
echo "== dag_concurrent_stress 60 s x 4 threads"; timeout -k 10 200 python tools/dag_concurrent_stress.py 60 4 2>&1 | tail -2
} 2>&1 | tee gpurun_out/r04_stress_end_of_round.log
echo "== workflow"; timeout -k 10 200 python tools/workflow_time.py 2>&1 | tee gpurun_out/r04_workflow_time.log | tail -12
