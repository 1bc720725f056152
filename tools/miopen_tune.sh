# One-off MIOpen find-db generation (exhaustive find on every conv shape of the step).  The result
# (detmatch_amd/miopen_db/*) is shipped in-tree; at run time MIOPEN_FIND_MODE=FAST only LOOKS UP the
# db and falls back to immediate mode for unknown shapes, so no run ever pays the search.
set -x
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH
export DM_CUDNN_BENCHMARK=1
python bench.py --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-140
DM_BENCH_WORKLOAD=pvrcnn python bench.py --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-140
ls -la $MIOPEN_USER_DB_PATH
du -sh $MIOPEN_USER_DB_PATH
# verification: lookup-only mode
export MIOPEN_FIND_MODE=FAST
time python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-140
