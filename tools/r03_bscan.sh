# batched single launch: the task list chosen by the launch's work (default) against the chain-bound list (CGP_DAG_BLAZY=0) and the
# work-bound one (=1), over batch shapes from a few small objects to the BASELINE configurations
S="2048:4:64 2048:8:64 2048:16:64 2048:32:64 3072:8:64 4096:2:64 4096:4:64 4096:8:64 4096:16:1024 6144:4:64 8192:2:64 8192:4:2500 8192:8:2500 2304:9:19 4096:8:1024 6144:8:64 8192:3:2500"
echo "== default (by the launch's work)"; timeout -k 5 300 python3 tools/batch_time.py "$S" | sed 's/steps.*//'
echo "== CGP_DAG_BLAZY=0 (window 2, lead 2, depth 8 / 16 / 32 by the tile count)"; timeout -k 5 300 python3 tools/batch_time.py "$S" CGP_DAG_BLAZY=0 | sed 's/steps.*//'
echo "== CGP_DAG_BLAZY=1 (window 1, lead 1, depth 16 / 24)"; timeout -k 5 300 python3 tools/batch_time.py "$S" CGP_DAG_BLAZY=1 | sed 's/steps.*//'
