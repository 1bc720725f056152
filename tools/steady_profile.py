"""Steady-state per-step kernel table from a rocprofv3 --kernel-trace CSV.

    python tools/steady_profile.py <..._kernel_trace.csv> --marker ema_f32 --steps 4 [--top 40]

A marker kernel that is launched exactly once per training step (the fused EMA for the DetMatch
workload; any per-step-unique kernel for others) delimits whole steps; everything between the
(steps+1)-th last and the last marker launch is aggregated, so warm-up (MIOpen find mode, lazy
allocations) never enters the table.  Output: per-kernel calls/step, us/step, share of the GPU-busy
time, plus busy vs wall per step (wall - busy = launch/host gaps).
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
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--skip-last', type=int, default=0,
                    help='iterations at the end of the trace to leave out (bench.py ends with two op-by-op measurement steps)')
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search(a.marker, r[2])]
    if a.skip_last:
        marks = marks[:-a.skip_last]
    if len(marks) < a.steps + 1:
        sys.exit('only %d marker launches' % len(marks))
    lo, hi = marks[-(a.steps + 1)], marks[-1]
    window = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / 1e3 / a.steps
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n in window:
        n = n.replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\((?!anonymous)[^()]*(\([^()]*\)[^()]*)*\)( const)?$', '', n)   # trailing parameter list
        n = n if len(n) <= 120 else n[:117] + '...'
        agg[n][0] += 1
        agg[n][1] += (e - s) / 1e3
    busy = sum(v[1] for v in agg.values()) / a.steps
    print('steps=%d  launches/step=%.0f  busy=%.1f us/step  wall=%.1f us/step  (gaps %.1f%%)' % (
        a.steps, len(window) / a.steps, busy, wall, 100 * (1 - busy / wall)))
    print('%10s %10s %7s  %s' % ('calls/step', 'us/step', '%busy', 'kernel'))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print('%10.1f %10.1f %6.1f%%  %s' % (c / a.steps, t / a.steps, 100 * t / a.steps / busy, n))


if __name__ == '__main__':
    main()
