#!/bin/bash
# LDS behaviour of the sparse-conv kernels of one layer: bash tools/pmc_spconv_lds.sh <out> [layer substring]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=${2:-subm3}
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/tools/run_layer.py $L 20 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/run_layer.py $L 20 > $O/p2.log 2>&1
cd $R; for p in p1 p2; do python3 tools/pmc_kernel_table.py $O/$p spconv > $O/$p.txt; rm -rf $O/$p; cat $O/$p.txt; done
