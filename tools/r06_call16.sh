#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_ssl_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for round in 1 2 3; do
  for v in "side_wgrad:A=1" "main_wgrad:DM_SIDE_WGRAD=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % (d['ms_per_step']))"
  done
done
