#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_ssl_gpu.py -q -m gpu -x -k "side_stream or key_point or lanes" 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
for round in 1 2 3; do
  for v in "wgrad_final_on_teacher_lane:A=1" "wgrad_on_side_stream:DM_WGRAD_TEACHER_LANE=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % (d['ms_per_step']))"
  done
done
