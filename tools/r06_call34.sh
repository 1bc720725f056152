#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_sched; mkdir -p $O
timeout 900 python -m pytest tests/test_ts_ssl_dataset.py tests/test_ssl_gpu.py tests/test_multirank_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee $O/tests.txt
export DM_BENCH_WATCHDOG=0
for round in 1 2; do timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  %.2f ms' % d['ms_per_step'], d['config'].get('ahead_of_previous_tail'))"; done | tee $O/bench.txt
