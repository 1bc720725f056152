#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/collect_profiles.sh trace pmc bench tools 2>&1 | tail -30
cd "${GRAFT_REPO_ROOT:-/root/repo}"
head -30 gpurun_out/r06/step_breakdown.txt
python - <<'PY'
import json
for n in ('bench_detmatch','bench_detmatch_mixed_precision','bench_waymo_fp32','bench_waymo_mixed_precision','bench_pvrcnn','bench_confthr'):
    try:
        d=json.loads(open('gpurun_out/r06/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['value'])
    except Exception as e: print(n,'FAILED',e)
PY
