#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|Error" | tail -8
for round in 1 2; do
  for v in "default:A=1" "pfe_on_main:DM_PFE_SIDE=0" "wgrad_on_main:DM_SIDE_WGRAD=0" "both_on_main:DM_PFE_SIDE=0 DM_SIDE_WGRAD=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % (d['ms_per_step']))"
  done
done
