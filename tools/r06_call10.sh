#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call10
mkdir -p $O
echo "== tests"
timeout 900 python -m pytest tests/test_blas_turn_gpu.py tests/test_fc_gemm_gpu.py -q -m gpu 2>&1 | tail -8
echo "== vendor GEMM census"
timeout 300 python tools/vendor_gemm_census.py 2>&1 | grep -v amdgpu.ids | tee $O/vendor_gemm_census.txt | cut -c1-260
echo "== bench (with the reference-compiled CPU baseline)"
timeout 600 python bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err; tail -3 $O/bench_detmatch.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call10/bench_detmatch.json').read().strip().splitlines()[-1])
c=d['cpu_baseline']
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'cpu_baseline', c['kind'], c['value'], c['cores'], c.get('pieces_1_thread'), c.get('all_threads'), 'port', c.get('port',{}).get('value'))
PY
