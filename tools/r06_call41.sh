#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export DM_LIB_PATH=$PWD/tools/altlib/lib_oldsgd.so; else unset DM_LIB_PATH; fi
    timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v sgd kernel  round $round  %.2f ms' % d['ms_per_step'])"
  done
done | tee gpurun_out/r06_sgd/ab.txt
