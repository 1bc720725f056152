R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_spconv16_gpu.py tests/test_pcdet_golden_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_spconv_layers.py --reps 40 > $O/spconv_layers2.txt 2>&1; grep -v "rulebook\|amdgpu" $O/spconv_layers2.txt | cut -c1-120
python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])"
