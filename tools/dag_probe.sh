#!/bin/bash
# ./tools/dag_probe.sh "<n>:<grid> ..." [levels]
for c in ${1:-256:2}; do
  n=${c%%:*}; g=${c##*:}
  for lvl in ${2:-0}; do
    echo "== n=$n grid=$g level $lvl"
    timeout -k 3 60 ./tools/dag_probe $n $lvl $g ${3:-1} ${4:-1}
    echo "rc=$?"
  done
done
