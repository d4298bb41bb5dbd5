#!/bin/bash
# ./tools/dag_sweep.sh <n> <tall> "<depth:window:dmax:fill%:lead:slope> ..."
for c in $3; do
  IFS=: read d w m f l sl <<< "$c"
  echo "== n=$1 tall=$2 depth=$d window=$w dmax=$m fill=$f lead=$l slope=$sl"
  timeout -k 3 60 ./tools/dag_probe $1 0 256 $2 $d $w $m $f ${l:-2} ${sl:-2} | grep -v "^launch"
done
