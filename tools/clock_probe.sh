#!/bin/bash
# How much of the dense-conv kernels' time is the clock: the same launches on random and on all-zero operands
# (MI355X_MICROARCH.md: the chip holds a higher clock when the matrix pipe toggles less).  bash tools/clock_probe.sh <out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in random zeros; do
  a=""; [ $m = zeros ] && a=zeros
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o kt -- python3 $R/tools/prof_dense_conv.py $a > /dev/null 2>&1
  f=$(find $O/kt_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m operands" >> $O/clock_probe.txt
  python3 - $f >> $O/clock_probe.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if n.startswith(('dconv_patch', 'dconv_wgrad9', 'dconv_gemm_bf16', 'dconv_wgrad_bf16')):
        print('%-50s calls %5s  mean %9.1f us' % (n[:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $O/kt_$m
done
cat $O/clock_probe.txt
