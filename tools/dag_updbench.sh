#!/bin/bash
# ./tools/dag_updbench.sh "<depth> ..."   : update-task benchmark of k_chol_dag (dbg level 9), n = 8192
for d in $1; do timeout -k 3 60 ./tools/dag_probe 8192 9 256 1 $d | grep updbench; done
