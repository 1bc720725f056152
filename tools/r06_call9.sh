#!/bin/bash
# Round 6, call 9: the FC GEMM: tests, per-layer timing against the vendor library and the conv GEMM, the step with it.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call9
mkdir -p $O
echo "== tests"
timeout 900 python -m pytest tests/test_fc_gemm_gpu.py tests/test_blas_turn_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== bench_fc"
timeout 600 python tools/bench_fc.py 2>&1 | grep -v amdgpu.ids | tee $O/fc_blas_vs_own.txt
echo "== bench A/B: FC GEMM on / off"
for round in 1 2; do
  for v in "fc_gemm:A=1" "vendor_fc:DM_FC_GEMM=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs DM_BENCH_WATCHDOG=0 timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $O/ab_${name}_$round.json 2> $O/ab_${name}_$round.err
    python - "$name" "$round" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r06_call9/ab_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
    print('%-12s round %s  %.2f ms/step' % (sys.argv[1], sys.argv[2], d['ms_per_step']))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'FAILED', e)
PY
  done
done
tail -3 $O/ab_fc_gemm_1.err
echo "== suite (ssl / chain / pvrcnn / frcnn)"
timeout 1200 python -m pytest tests/test_ssl_gpu.py tests/test_pvrcnn_gpu.py tests/test_frcnn_gpu.py tests/test_pcdet_golden_gpu.py tests/test_fused_end_to_end_gpu.py -x -q -m gpu 2>&1 | tail -6
