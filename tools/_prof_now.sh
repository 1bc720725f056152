R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/err.txt
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 70 > $O/steady.txt
python3 $R/tools/dconv_calls.py $f > $O/dconv_calls.txt
rm -rf $O/kt
head -9 $O/steady.txt | cut -c1-100; head -12 $O/dconv_calls.txt | cut -c1-110
cd $R; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
