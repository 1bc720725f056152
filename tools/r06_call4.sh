#!/bin/bash
# Round 6, call 4: new GPU tests, full GPU suite, why one rank under torchrun is 28 ms slower, a first bench line, sync census.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call4
mkdir -p $O
echo "== new tests"
timeout 900 python -m pytest tests/test_glue_lazy_gpu.py tests/test_blas_turn_gpu.py tests/test_ssl_match_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== torchrun A/B (100 iterations each, three lanes)"
timeout 200 python tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/plain: /'
OMP_NUM_THREADS=1 timeout 200 python tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/OMP_NUM_THREADS=1: /'
timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/torchrun nccl: /'
OMP_NUM_THREADS=8 timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/torchrun nccl OMP=8: /'
DM_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/torchrun gloo: /'
DM_TWO_LANES=0 timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok" | sed 's/^/torchrun nccl one lane: /'
echo "== bench"
timeout 600 python bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err; tail -c 600 $O/bench_detmatch.json | head -c 600; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call4/bench_detmatch.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', {k: d['roofline'][k] for k in ('bound','achieved','frac','avg_us','launches')}, d['roofline']['mfma'], d['roofline']['timed_region_event_pairs'])
PY
echo "== sync census"
timeout 300 python tools/find_syncs.py detmatch > $O/host_syncs.txt 2>&1; head -30 $O/host_syncs.txt
echo "== full GPU suite"
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
