R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/err.txt
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); s=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $s $O/kernel_stats.csv
python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 70 > $O/steady.txt
rm -rf $O/kt
cd $R; python -m pytest tests/test_ssl_gpu.py -m gpu -q 2>&1 | tail -3
head -75 $O/steady.txt | cut -c1-130
