#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_sort_rows_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -6
for round in 1 2; do DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench round $round  %.2f ms/step' % (d['ms_per_step']))"; done
timeout 400 python tools/lane_soak.py run 1500 2>&1 | grep "steps ok"
