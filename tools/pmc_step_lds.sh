#!/bin/bash
# LDS bank conflicts of every kernel of the DetMatch step, ranked by conflict cycles: bash tools/pmc_step_lds.sh <out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/p1.log 2>&1
cd $R; python3 - $O/p1 > $O/lds_conflicts.txt <<'PY'
import collections, csv, glob, os, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'^void ', '', n).split('(')[0][:70]
        tot[n][r['Counter_Name']] += float(r['Counter_Value'])
rows = sorted(tot.items(), key=lambda kv: -kv[1]['SQ_LDS_BANK_CONFLICT'])
print('%-70s %14s %14s %6s %14s' % ('kernel (4 steps)', 'conflict cyc', 'LDS cyc', 'frac', 'busy cyc'))
for n, c in rows[:40]:
    a = c['SQ_LDS_IDX_ACTIVE']
    print('%-70s %14.0f %14.0f %6.2f %14.0f' % (n, c['SQ_LDS_BANK_CONFLICT'], a, c['SQ_LDS_BANK_CONFLICT'] / a if a else 0, c['SQ_BUSY_CYCLES']))
PY
rm -rf $O/p1; cat $O/lds_conflicts.txt
