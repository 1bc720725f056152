#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call12
mkdir -p $O
timeout 600 python -m pytest tests/test_fc_gemm_gpu.py -q -m gpu 2>&1 | tail -3
for v in 321 322 641 642; do
  echo "== DM_FC_VARIANT=$v"
  DM_FC_VARIANT=$v timeout 300 python -m pytest tests/test_fc_gemm_gpu.py -q -m gpu -x 2>&1 | tail -1
  DM_FC_VARIANT=$v timeout 600 python tools/bench_fc.py 2>&1 | grep -v amdgpu.ids | cut -c1-170 | tee $O/fc_variant_$v.txt
done
