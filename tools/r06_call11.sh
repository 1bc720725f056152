#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call11
mkdir -p $O
timeout 600 python -m pytest tests/test_fc_gemm_gpu.py -q -m gpu 2>&1 | tail -3
timeout 600 python tools/bench_fc.py 2>&1 | grep -v amdgpu.ids | tee $O/fc_blas_vs_own.txt
for i in 1 2; do DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench %.2f ms/step' % d['ms_per_step'])"; done
