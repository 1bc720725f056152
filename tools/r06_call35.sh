#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_bigtile; mkdir -p $O
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do
  for v in 512 256 128; do
    DM_DCONV_BIG_TILE_MIN=$v timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('big tile from $v tiles  round $round  %.2f ms' % d['ms_per_step'])"
  done
done | tee $O/ab.txt
