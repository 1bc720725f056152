#!/bin/bash
# L1 / L2 behaviour of the dominant sparse kernel (subm3 64 -> 64 forward, 20 launches): separate counter passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/p1 -- python3 $R/tools/run_layer.py subm3 20 > $O/p1.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr TCC_TAG_STALL_sum --output-format csv -d $O/p2 -- python3 $R/tools/run_layer.py subm3 20 > $O/p2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/run_layer.py subm3 20 > $O/p3.log 2>&1
cd $R; for p in p1 p2 p3; do python3 tools/pmc_kernel_table.py $O/$p spconv_gr; rm -rf $O/$p; done
