R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O; cd $R
for i in 1 2 3; do
DM_GC_PAUSE=0 python3 tools/phase_timeline.py 2>&1 | grep "host issued" | sed 's/^/gc-auto  /'
DM_GC_PAUSE=1 python3 tools/phase_timeline.py 2>&1 | grep "host issued" | sed 's/^/gc-pause /'
done
