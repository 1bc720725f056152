#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_t2dfirst; mkdir -p $O
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  %.2f ms' % d['ms_per_step'])"; done | tee $O/bench.txt
timeout 100 python3 tools/steady_timeline.py 7 2>&1 | grep -v "amdgpu.ids" > $O/steady.txt
timeout 900 python -m pytest tests/test_ssl_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee $O/tests.txt
