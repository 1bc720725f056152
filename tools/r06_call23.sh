#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_gantt
mkdir -p $O
cd /tmp
export DM_BENCH_WATCHDOG=0
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --steps 12 --warmup 4 > $O/bench.json 2> $O/bench.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/stream_gantt.py $f > $O/stream_gantt.txt 2>&1
rm -rf $O/kt
cat $O/stream_gantt.txt | head -150
