#!/bin/bash
mkdir -p gpurun_out
echo "== probe"; (timeout -k 5 60 bash tools/dag_probe.sh "512:4 4096:256" 0; DAG_PROBE_STEPS=1 timeout -k 5 60 bash tools/dag_probe.sh "8192:256" 0) 2>&1 | tee gpurun_out/r04_dag_phase_profile_b.log | grep -E "leaf lane|max|done|rc="
echo "== dag tests"; timeout -k 10 600 python -m pytest tests/test_dag.py tests/test_round2_gpu.py -m gpu -x -q > gpurun_out/r04_tests6.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04_tests6.log
echo "== chol rate"; python tools/chol_rate.py "2048 4096 8192 16384" 2>&1 | tee gpurun_out/r04_chol_rate_b.log
