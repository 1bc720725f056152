#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_trunk
mkdir -p $O
cd $R
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do
  for v in "default:A=1" "trunk_on_2d_lane:DM_TRUNK_ON_2D_LANE=1" "trunk_2d+sup_per_lane:DM_TRUNK_ON_2D_LANE=1 DM_SUP_BWD_PER_LANE=1"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s round $round  %.2f ms' % ('$n', d['ms_per_step']))" || echo "$n round $round FAILED"
  done
done 2>&1 | tee $O/ab.txt
