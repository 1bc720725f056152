#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_cumask; mkdir -p $O
export DM_BENCH_WATCHDOG=0
for round in 1 2; do
  for v in "default:A=1" "lane1_75pct:DM_LANE1_CU_MASK=7" "lane1_50pct:DM_LANE1_CU_MASK=5" "lane1+2_75pct:DM_LANE1_CU_MASK=7 DM_LANE2_CU_MASK=7" "lane1_75_lane2_other:DM_LANE1_CU_MASK=7 DM_LANE2_CU_MASK=e"; do
    n=${v%%:*}; e=${v#*:}
    env $e timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>$O/err_$n.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s round $round  %.2f ms' % ('$n', d['ms_per_step']))" || { echo "$n round $round FAILED"; tail -3 $O/err_$n.txt; }
  done
done 2>&1 | tee $O/ab.txt
