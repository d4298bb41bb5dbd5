#!/bin/bash
# ./tools/dag_probe.sh "<n>:<grid> ..." [levels]
for c in ${1:-256:2}; do
  n=${c%%:*}; g=${c##*:}
  for lvl in ${2:-0}; do
    echo "== n=$n grid=$g level $lvl"
    timeout -k 3 40 ./tools/dag_probe $n $lvl $g
    echo "rc=$?"
  done
done
