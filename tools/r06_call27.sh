#!/bin/bash
# teacher-ahead ordering: its tests, a 1500-iteration soak, a 1000-iteration soak under RCCL (one rank), then the whole GPU suite
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_ahead4; mkdir -p $O
timeout 900 python -m pytest tests/test_ssl_gpu.py -q -m gpu -x -k "ahead or supervised_2d" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee $O/new_tests.txt
timeout 400 python tools/lane_soak.py run 1500 2>&1 | grep -E "steps ok|Error|Traceback" | tee $O/soak.txt
timeout 400 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 1000 2>&1 | grep -E "steps ok|process group|Error|Traceback" | tee $O/soak_nccl.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -6 | tee $O/suite.txt
