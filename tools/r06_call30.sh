#!/bin/bash
# long soak of the final order: 10 000 iterations under torchrun / nccl (one rank), heartbeat-watched
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_soak; mkdir -p $O
timeout 1500 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 10000 2>&1 | grep -E "steps ok|process group|Error|Traceback|step 9999|step 4999" | tee $O/soak_nccl_10000.txt
