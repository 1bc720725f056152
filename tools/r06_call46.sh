#!/bin/bash
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/r06_grg; mkdir -p $O
cd $R
for v in 128 256 1024 4096; do DM_GRG_WGS=$v timeout -k 5 100 python3 tools/bench_group_grad.py 2>&1 < /dev/null | tail -2; done | tee $O/micro.txt
cd /tmp
for v in 256 4096; do
  DM_GRG_WGS=$v timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/p$v -- python3 $R/tools/bench_group_grad.py > $O/p$v.log 2>&1 < /dev/null
  (cd $R; python3 tools/pmc_kernel_table.py $O/p$v group_rows_grad 2>/dev/null | head -12) | tee $O/pmc_$v.txt
  rm -rf $O/p$v
done
