#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call14
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
for i in 1 2 3; do DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench %.2f ms/step' % d['ms_per_step'])"; done
timeout 200 python tools/phase_timeline.py 2>&1 | grep -v amdgpu.ids > $O/phase_timeline.txt; grep -E "ema|backward\+clip|host issued" $O/phase_timeline.txt
