R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R
python -m pytest tests/test_dense_conv_gpu.py -m gpu -q 2>&1 | tail -1
(cd tools && python3 bench_dense_conv_math.py 2>&1 | grep -v "amdgpu.ids" > $O/math_modes5.txt); cat $O/math_modes5.txt | cut -c1-130
