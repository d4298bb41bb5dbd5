#!/bin/bash
mkdir -p gpurun_out
echo "== full gpu suite"; timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r04_tests_final_a.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04_tests_final_a.log
echo "== final pass 2 (rocprofv3 kernel statistics)"; bash tools/final_pass_r04.sh 2
