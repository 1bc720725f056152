R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt; grep -n "FAILED\|^E " $O/tests.txt | head -20
