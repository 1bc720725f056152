#!/bin/bash
# final state of round 6: whole GPU suite, smoke, profiles + bench family, driver-style bench
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== full GPU suite"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh trace pmc bench tools 2>&1 | tail -3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06/bench_driver_style.json
python - <<'PY'
import json
for n in ('bench_detmatch','bench_driver_style','bench_detmatch_mixed_precision','bench_waymo_fp32','bench_waymo_mixed_precision','bench_pvrcnn','bench_confthr'):
    try:
        d=json.loads(open('gpurun_out/r06/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['value'], d['roofline'].get('frac'), d['roofline'].get('avg_us'))
    except Exception as e: print(n,'FAILED',e)
PY
grep "spconv_gr<64, 64>" gpurun_out/r06/detmatch_bench_kernel_stats.csv | cut -d, -f2-4
