#!/bin/bash
# ./tools/dag_sweep.sh <n> <tall> "<depth:window:deep_min:fill%:lead> ..."
for c in $3; do
  IFS=: read d w m f l <<< "$c"
  echo "== n=$1 tall=$2 depth=$d window=$w deep_min=$m fill=$f lead=$l"
  timeout -k 3 60 ./tools/dag_probe $1 0 256 $2 $d $w $m $f ${l:-3} | grep -v "^launch\|^done"
done
