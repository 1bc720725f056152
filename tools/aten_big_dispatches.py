"""The largest individual ATen / runtime copy-fill dispatches of one steady iteration (rocprofv3 kernel trace): what a
fused kernel would have to replace to matter.

    python tools/aten_big_dispatches.py <..._kernel_trace.csv> --marker ema_f32 [--top 40]
"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--marker', required=True)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--skip-last', type=int, default=0,
                    help='iterations at the end of the trace to leave out (bench.py ends with two op-by-op measurement steps)')
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            grid = r.get('Grid_Size') or r.get('Grid_Size_X') or '?'   # threads
            grid = '%sx%s' % (grid, r.get('Workgroup_Size_X', '?'))
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], grid, r.get('Queue_Id', '?')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search(a.marker, r[2])]
    if a.skip_last:
        marks = marks[:-a.skip_last]
    if len(marks) < 2:
        sys.exit('only %d marker launches' % len(marks))
    window = rows[marks[-2]:marks[-1]]
    pick = [r for r in window if re.search(r'at::native|rocclr|rocprim|hipcub', r[2])]
    total = sum(e - s for s, e, *_ in pick) / 1e3
    print('one iteration: %d ATen / runtime / rocprim dispatches, %.1f us' % (len(pick), total))
    hist = defaultdict(lambda: [0, 0.0])
    for s, e, n, g, q in pick:
        d = (e - s) / 1e3
        b = '<5' if d < 5 else '5-10' if d < 10 else '10-20' if d < 20 else '20-50' if d < 50 else '50-100' if d < 100 else '>=100'
        hist[b][0] += 1
        hist[b][1] += d
    print('by duration (us): ' + ', '.join('%s: %d x = %.0f us' % (b, hist[b][0], hist[b][1])
                                           for b in ('<5', '5-10', '10-20', '20-50', '50-100', '>=100') if b in hist))
    prev_name = {}
    for i, r in enumerate(window):
        prev_name[id(r)] = window[i - 1][2] if i else ''
    print('%9s %12s %5s  %-90s | previous kernel on the device' % ('us', 'grid', 'queue', 'kernel'))
    for r in sorted(pick, key=lambda r: r[0] - r[1])[:a.top]:
        s, e, n, g, q = r
        short = re.sub(r'\s+', ' ', n)[:90]
        print('%9.1f %12s %5s  %-90s | %s' % ((e - s) / 1e3, g, q, short, re.sub(r'\(.*', '', prev_name[id(r)])[-60:]))


if __name__ == '__main__':
    main()
