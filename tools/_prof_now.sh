R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_spconv_gpu.py -m gpu -q -x -k "properties_full_size" 2>&1 | tail -15
