#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pmc1x1; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python3 $R/tools/prof_conv1x1.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/p2 -- python3 $R/tools/prof_conv1x1.py > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/prof_conv1x1.py > $O/kt.log 2>&1
cd $R; python3 tools/pmc_kernel_table.py $O/p1 dconv > $O/t1.txt; python3 tools/pmc_kernel_table.py $O/p2 dconv > $O/t2.txt
s=$(find $O/kt -name "*kernel_stats.csv" | head -1); grep -i "dconv" $s | cut -c1-60,230-330 > $O/stats.txt
rm -rf $O/p1 $O/p2 $O/kt; cat $O/t1.txt $O/t2.txt $O/stats.txt | cut -c1-260 | head -60
