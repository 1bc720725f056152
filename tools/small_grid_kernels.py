"""Kernels of a rocprofv3 --kernel-trace CSV that run long on few workgroups (candidates for 'more workgroups / more bytes in
flight'): per kernel name and grid size, calls, mean duration, workgroups.
    python tools/small_grid_kernels.py <kernel_trace.csv> [min_us] [max_workgroups]"""
import csv
import re
import sys
from collections import defaultdict

f = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
max_wg = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
agg = defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for r in csv.DictReader(fh):
        gs = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        ws = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        wgs = gs // max(ws, 1)
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'\(.*', '', name)[:64]
        a = agg[(name, wgs, ws)]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = [(t / c, c, k) for k, (c, t) in agg.items() if t / c >= min_us and k[1] <= max_wg]
rows.sort(reverse=True)
print('%-66s %8s %6s %7s %9s' % ('kernel', 'workgrps', 'wgsize', 'calls', 'mean us'))
for mean, c, (name, wgs, ws) in rows[:60]:
    print('%-66s %8d %6d %7d %9.1f' % (name, wgs, ws, c, mean))
