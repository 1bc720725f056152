#!/bin/bash
# per-kernel durations of the dense-conv forward layers (rocprofv3 kernel trace): bash tools/prof_conv_layers.sh <out> [substring]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/run_conv_layer.py "$2" 10 > /dev/null 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size','?'), r.get('Workgroup_Size','?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
agg=collections.OrderedDict()
for s,e,n,g,w in rows:
    if 'dconv' not in n: continue
    n=n.replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    k=(n,g)
    a=agg.setdefault(k,[0,0.0,1e18]); a[0]+=1; a[1]+=(e-s)/1e3; a[2]=min(a[2],(e-s)/1e3)
for (n,g),(c,t,mn) in agg.items():
    print('%-58s grid %-10s calls %3d  mean %8.1f us  min %8.1f us' % (n[:58], g, c, t/c, mn))
PY
rm -rf $O/kt
