#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_small; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 > $O/b.json 2>/dev/null < /dev/null
f=$(find $O/kt -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -1 "$f" | cut -c1-400; python3 $R/tools/small_grid_kernels.py "$f" 40 1100 > $O/small_grid.txt 2>&1; fi
rm -rf $O/kt
cat $O/small_grid.txt | head -50
