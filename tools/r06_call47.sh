#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_grg; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_ssl_gpu.py -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed|^FAILED|rror" | tail -3 | tee $O/tests.txt
export DM_BENCH_WATCHDOG=0
for round in 1 2 3 4; do
  for v in 4096 256; do
    DM_GRG_WGS=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('DM_GRG_WGS=$v round $round  %.2f ms' % d['ms_per_step'])"
  done
done | tee $O/ab2.txt
