#!/bin/bash
# ./tools/r03_pmc.sh <tag> : SQ counters + HBM traffic of the default bench command (k_chol_dag, 8 objects per launch, >= 3
# eight-object launches per pass), separate rocprofv3 --pmc passes -> gpurun_out/<tag>_*.  Run on the GPU box from the repo root.
tag=${1:-r03}
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out
args="--no-extras --no-cpu-baseline --steps 3 --warmup 0"
# SQ passes (8 SQ slots per pass)
(cd /tmp && rm -rf /tmp/pmc_sq1 && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/pmc_sq1 -o run --output-format csv -- python3 $root/bench.py $args > /tmp/pmc_sq1.log 2>&1) || { tail -5 /tmp/pmc_sq1.log; exit 1; }
python3 tools/pmc_sq.py $(find /tmp/pmc_sq1 -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_pmc_sq_a.csv
echo "sq pass a done"
(cd /tmp && rm -rf /tmp/pmc_sq2 && rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d /tmp/pmc_sq2 -o run --output-format csv -- python3 $root/bench.py $args > /tmp/pmc_sq2.log 2>&1) || { tail -5 /tmp/pmc_sq2.log; exit 1; }
python3 tools/pmc_sq.py $(find /tmp/pmc_sq2 -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_pmc_sq_b.csv
echo "sq pass b done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o run --output-format csv -- python3 $root/bench.py $args > /tmp/pmc_$c.log 2>&1) || { tail -5 /tmp/pmc_$c.log; exit 1; }
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $f $w 8 gpurun_out/${tag}_pmc_traffic_c3_dagbatch.json k_chol_dag
echo "traffic done"
