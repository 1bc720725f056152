#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_prio
mkdir -p $O
cd $R
export DM_BENCH_WATCHDOG=0
python3 - <<'PY'
import torch
for p in (-2,-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print('asked', p, 'got', s.priority)
PY
for round in 1 2; do
  for v in "default:A=1" "lanes_low:DM_LANE_PRIORITY=1" "lanes_aux_low:DM_LANE_PRIORITY=1 DM_AUX_PRIORITY=1" "main_high:DM_MAIN_PRIORITY=-1" "main_high_aux_high:DM_MAIN_PRIORITY=-1 DM_AUX_PRIORITY=-1"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s round $round  %.2f ms' % ('$n', d['ms_per_step']))" || echo "$n round $round FAILED"
  done
done 2>&1 | tee $O/ab.txt
