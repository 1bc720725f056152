#!/bin/bash
# kernel durations + SQ counters of the dense weight-gradient kernels: bash tools/pmc_wgrad.sh <out> <layer substring> [mode]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
M=${3:-fp32_split}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/run_wgrad_layer.py "$2" 10 $M > /dev/null 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); grep -i "dconv" $f | cut -c1-200 > $O/kernels_$M.txt; rm -rf $O/kt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python3 $R/tools/run_wgrad_layer.py "$2" 3 $M > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/p2 -- python3 $R/tools/run_wgrad_layer.py "$2" 3 $M > $O/p2.log 2>&1
cd $R; python3 tools/pmc_kernel_table.py $O/p1 dconv > $O/t1_$M.txt; python3 tools/pmc_kernel_table.py $O/p2 dconv > $O/t2_$M.txt; rm -rf $O/p1 $O/p2
cat $O/kernels_$M.txt $O/t1_$M.txt $O/t2_$M.txt
