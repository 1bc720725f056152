#!/bin/bash
# How much of the dense-conv kernels' time is the clock: the same launches on random and on all-zero operands
# (MI355X_MICROARCH.md: the chip holds a higher clock when the matrix pipe toggles less).  bash tools/clock_probe.sh <out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in random zeros; do
  a=""; [ $m = zeros ] && a=zeros
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o kt -- python3 $R/tools/prof_dense_conv.py $a > /dev/null 2>&1
  f=$(find $O/kt_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m operands" >> $O/clock_probe.txt
  grep "dconv_patch\|dconv_wgrad9\|dconv_gemm_bf16" $f | awk -F'","' '{gsub(/"/,"",$1); n=$1; sub(/\(.*/,"",n); printf "%-60s calls %5s  mean %10.1f us\n", substr(n,1,60), $2, $4/1000}' >> $O/clock_probe.txt
  rm -rf $O/kt_$m
done
cat $O/clock_probe.txt
