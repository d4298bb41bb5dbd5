# n dbg grid tall depth window dmax fill lead slope
for cfg in "8 4 8 70 2 1" "8 3 8 70 2 1" "8 2 8 70 2 1" "8 2 8 70 1 1" "8 3 8 70 1 1" "8 4 8 70 1 1" "8 4 8 50 2 1" "8 4 8 100 2 1" "8 4 12 70 2 1" "12 4 12 70 2 1" "8 4 8 70 2 2" "6 4 6 70 2 1" "8 3 8 100 1 1" "8 2 8 100 2 1"; do
  set -- $cfg
  r=$(timeout -k 3 40 ./tools/dag_probe 8192 0 256 1 $1 $2 $3 $4 $5 $6 | grep "done in\|max |L" | tr '\n' ' ' | sed 's/(host wall).*max/ max/')
  echo "depth $1 window $2 dmax $3 fill $4 lead $5 slope $6: $r"
done
