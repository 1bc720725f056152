#!/bin/bash
# Round 6, call 6: deferred rulebooks + off-main-lane FCs: tests, bench, timeline, syncs; nccl one-rank timing at 4 queues.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_call6
mkdir -p $O
echo "== targeted tests"
timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_blas_turn_gpu.py tests/test_fps_batch_gpu.py tests/test_glue_lazy_gpu.py -x -q -m gpu 2>&1 | tail -12
echo "== bench"
timeout 600 python bench.py > $O/bench_detmatch.json 2> $O/bench_detmatch.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_call6/bench_detmatch.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', {k: d['roofline'][k] for k in ('bound','achieved','frac','avg_us','launches')}, d.get('note'))
PY
tail -3 $O/bench_detmatch.err
echo "== timeline"
timeout 300 python tools/phase_timeline.py > $O/phase_timeline.txt 2>&1; head -42 $O/phase_timeline.txt
echo "== sync census"
timeout 300 python tools/find_syncs.py detmatch > $O/host_syncs.txt 2>&1; head -12 $O/host_syncs.txt
echo "== nccl one rank (4 queues pinned by the tool)"
timeout 300 python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nproc-per-node 1 tools/lane_soak.py run 100 2>&1 | grep "steps ok"
timeout 200 python tools/lane_soak.py run 100 2>&1 | grep "steps ok"
echo "== full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
