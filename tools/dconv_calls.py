"""Per-launch-shape table of the dense-conv kernels (or of the kernels matching the regular expression
given as second argument) from a rocprofv3 --kernel-trace CSV: kernel, workgroups, calls, mean / total
duration (steady window delimited by the once-per-step EMA kernel)."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r.get('Grid_Size_Y', 1) or 1)))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'ema_f32' in r[2]]
if '--skip-last' in sys.argv:          # bench.py ends with two op-by-op measurement steps: leave them out
    k = sys.argv.index('--skip-last')
    marks = marks[:-int(sys.argv[k + 1])]
    del sys.argv[k:k + 2]
steps = 4
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else 'dconv')
lo, hi = marks[-(steps + 1)], marks[-1]
agg = defaultdict(lambda: [0, 0.0])
for s, e, n, gx, gy in rows[lo:hi]:
    if not pat.search(n):
        continue
    n = re.sub(r'\(.*', '', n.replace('void ', '').replace('(anonymous namespace)::', ''))
    a = agg[(n, gx, gy)]
    a[0] += 1
    a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values()) / steps
print('%s kernels: %.1f us/step' % ('dense conv' if len(sys.argv) <= 2 else sys.argv[2], tot))
print('%-52s %8s %5s %8s %9s %9s' % ('kernel', 'blocks', 'y', 'calls', 'mean us', 'us/step'))
for (n, gx, gy), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-52s %8d %5d %8.1f %9.1f %9.1f' % (n[:52], gx, gy, c / steps, t / c, t / steps))
