R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R
python -m pytest tests/test_dense_conv_gpu.py -m gpu -q 2>&1 | tail -1
(cd tools && python3 bench_dense_conv_math.py 2>&1 | grep -v "amdgpu.ids" > $O/math_modes6.txt); cat $O/math_modes6.txt | cut -c1-130
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
