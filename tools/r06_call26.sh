#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_ahead3
mkdir -p $O
cd $R
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do
  for v in 1 0; do
    DM_SUP_BWD_PER_LANE=$v timeout 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 > $O/b_${v}_$round.json 2> $O/b_${v}_$round.err
    python3 - <<PY
import json
try:
    d=json.loads(open('$O/b_${v}_$round.json').read().strip().splitlines()[-1]); print('sup_per_lane=$v round $round: %.2f ms' % d['ms_per_step'])
except Exception as e: print('sup_per_lane=$v round $round FAILED', e)
PY
  done
done 2>&1 | tee $O/ab.txt
timeout 300 python3 tools/steady_timeline.py 7 > $O/steady_1.txt 2> $O/steady_1.err
timeout 900 python3 -m pytest tests/test_ssl_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 | tee $O/ssl_tests.txt
