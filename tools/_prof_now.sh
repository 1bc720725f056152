R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; grep -n "FAILED\|^E " $O/tests.txt | head -30
python3 tools/phase_timeline.py 2>&1 | grep -v "amdgpu.ids" > $O/phase_timeline.txt; head -40 $O/phase_timeline.txt
