R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2 3; do
  for v in "fp32:A=1" "mixed:DM_CONV_MATH=bf16"; do
    n=${v%%:*}; e=${v#*:}
    env $e python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %d %.2f %s' % ('$n', $i, d['ms_per_step'], d['dtype']))"
  done
done
cd /tmp; export TMPDIR=/tmp
DM_CONV_MATH=bf16 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/b.json 2> $O/err.txt
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/steady_profile.py $f --marker ema_f32 --steps 8 --top 30 > $O/steady_mixed.txt
rm -rf $O/kt; head -24 $O/steady_mixed.txt | cut -c1-120
