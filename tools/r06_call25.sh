#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_ahead
mkdir -p $O
cd $R
for v in 0 1; do
  DM_TEACHER_AHEAD=$v timeout 300 python3 tools/steady_timeline.py 7 > $O/steady_$v.txt 2> $O/steady_$v.err
  tail -3 $O/steady_$v.err
done
