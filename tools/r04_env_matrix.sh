#!/bin/bash
# the GPU suite under the round's switches, one at a time (every opt-in / opt-out route must pass the same parity tests)
mkdir -p gpurun_out
L=gpurun_out/r04_env_matrix.log; : > $L
for e in ${CGP_MATRIX:-"CGP_NO_POTRF4=1" "CGP_POTRF4_MIN=2" "CGP_NO_XCD=1" "CGP_NO_PACK=1" "CGP_PIPE_CHUNK=0" "CGP_PIPE_CHUNK=3" "CGP_LL=1"}; do
  echo "== $e" | tee -a $L
  env $e timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee -a $L
done
