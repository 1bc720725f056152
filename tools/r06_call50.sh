#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
timeout -k 10 900 python3 -m pytest tests/test_pcdet_golden_gpu.py tests/test_ssl_gpu.py tests/test_pvrcnn_gpu.py -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed|^FAILED|rror" | tail -4
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  %.2f ms' % d['ms_per_step'])"; done
