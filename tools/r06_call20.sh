#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do
  for v in "default:A=1" "lookahead:DM_LOOKAHEAD=1" "teacher_pfe_side:DM_PFE_SIDE_EVAL=1"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name round $round  %.2f ms/step' % (d['ms_per_step']))"
  done
done
timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 300 2>&1 | grep "steps ok\|process group"
