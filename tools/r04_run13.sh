#!/bin/bash
mkdir -p gpurun_out
bash tools/final_pass_r04.sh 3 2>&1 | grep -E "done"
