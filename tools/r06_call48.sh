#!/bin/bash
# final state: whole GPU suite, smoke, the judged profile set and five repeat bench lines
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
echo "== full GPU suite"
timeout -k 10 900 python3 -m pytest tests -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed|^FAILED" | tail -6
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
bash tools/collect_profiles.sh trace pmc bench tools < /dev/null 2>&1 | tail -2
cd $R
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > gpurun_out/r06/bench_driver_style.json
for i in 1 2 3 4 5; do DM_BENCH_WATCHDOG=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i: %.2f ms/step = %.2f it/s, roofline kernel %.2f us (frac %.3f)' % (d['ms_per_step'], d['value'], d['roofline']['avg_us'], d['roofline']['frac']))"; done | tee gpurun_out/r06/bench_repeat.txt
python3 - <<'PY'
import json, csv
for n in ('bench_detmatch','bench_driver_style','bench_detmatch_mixed_precision','bench_waymo_fp32','bench_waymo_mixed_precision','bench_pvrcnn','bench_confthr'):
    try:
        d=json.loads(open('gpurun_out/r06/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['value'], d['roofline'].get('frac'), d['roofline'].get('avg_us'))
    except Exception as e: print(n,'FAILED',e)
for row in csv.DictReader(open('gpurun_out/r06/detmatch_bench_kernel_stats.csv')):
    if 'spconv_gr<64, 64>' in row['Name'] or 'group_rows_grad_combine' in row['Name']: print(row['Name'][:60], row['Calls'], row['AverageNs'])
PY
