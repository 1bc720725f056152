#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> steady table, category breakdown, dense-conv shapes.
#   bash tools/prof_step.sh <out dir under gpurun_out>   (extra environment is inherited by the bench)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
s=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $s $O/kernel_stats.csv
python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 400 > $O/step_steady.txt
python3 $R/tools/step_breakdown.py $f --steps 8 > $O/step_breakdown.txt
python3 $R/tools/dconv_calls.py $f > $O/dense_conv_launch_shapes.txt
rm -rf $O/kt
cat $O/step_breakdown.txt
