#!/bin/bash
# ./tools/dag_batch.sh <n> <tall> "<nmat> ..." [depth:window:dmax:fill:lead:slope]
cfg=${4:-8:4:8:70:2:1}
IFS=: read d w m f l sl <<< "$cfg"
for b in $3; do
  echo "== n=$1 tall=$2 objects=$b schedule=$cfg"
  DAG_PROBE_NMAT=$b DAG_PROBE_SCHED_WORKERS=${SCHED_WORKERS:-255} timeout -k 3 120 ./tools/dag_probe $1 0 256 $2 $d $w $m $f $l $sl | grep -v "^launch"
done
