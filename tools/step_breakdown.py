"""Where a DetMatch iteration's device time goes, from a rocprofv3 --kernel-trace CSV: kernel time by category, the
union of busy intervals (the device runs SOMETHING), busy time per HIP queue, and the idle gaps of the union by size.

    python tools/step_breakdown.py <..._kernel_trace.csv> [--marker ema_f32] [--steps 8]
"""
import argparse
import csv
import re
from collections import defaultdict

CATS = [
    ('dense conv', r'dconv_'),
    ('fps', r'fps_kernel'),
    ('sparse conv + rulebook', r'spconv|rb_|rulebook|tile_order|pack_rows|pack_weights|pairs_|voxel|hash_'),
    ('batch norm', r'bn_'),
    ('row gemm', r'rowgemm'),
    ('blas (Tensile)', r'Cijk_'),
    ('ball query / grouping', r'ball_query|group_|query_group'),
    ('roi align', r'roi_align'),
    ('nms / iou', r'nms_|iou|overlap'),
    ('sort / scan (rocprim)', r'rocprim|hipcub'),
    ('aten element-wise / copy / fill', r'at::native|rocclr|Memcpy|Memset'),
    ('optimizer / ema', r'adamw|sgd_|ema_'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--marker', default='ema_f32')
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--skip-last', type=int, default=0,
                    help='iterations at the end of the trace to leave out (bench.py ends with two op-by-op measurement steps)')
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                         r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if re.search(a.marker, r[2])]
    if a.skip_last:
        marks = marks[:-a.skip_last]
    steps = min(a.steps, len(marks) - 1)
    lo, hi = marks[-(steps + 1)], marks[-1]
    win = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / 1e3 / steps
    cat_t, cat_n = defaultdict(float), defaultdict(int)
    per_q = defaultdict(float)
    for s, e, n, q, st in win:
        for name, pat in CATS:
            if re.search(pat, n):
                break
        else:
            name = 'other own kernels'
        cat_t[name] += (e - s) / 1e3 / steps
        cat_n[name] += 1
        per_q[(q, st)] += (e - s) / 1e3 / steps
    # union of busy intervals
    busy, gaps, cur_s, cur_e = 0.0, [], win[0][0], win[0][1]
    for s, e, *_ in win[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('steps=%d  wall %.1f ms/step  launches %.0f/step  sum of kernel time %.1f ms  union busy %.1f ms  idle %.1f ms'
          % (steps, wall / 1e3, len(win) / steps, sum(cat_t.values()) / 1e3, busy / 1e6 / steps, (wall - busy / 1e3 / steps) / 1e3))
    print('%-36s %10s %10s' % ('category', 'ms/step', 'launches'))
    for name, t in sorted(cat_t.items(), key=lambda kv: -kv[1]):
        print('%-36s %10.2f %10.0f' % (name, t / 1e3, cat_n[name] / steps))
    print('busy per (queue, stream): ' + ', '.join('%s/%s %.1f ms' % (q, st, t / 1e3) for (q, st), t in
                                                   sorted(per_q.items(), key=lambda kv: -kv[1])[:8]))
    # the two phases of an iteration on the main queue: [optimizer of the previous step .. EMA] = the forward passes
    # with their early backward passes; [EMA .. optimizer] = the final backward (deferred 2D trunk included)
    opt = [i for i, r in enumerate(rows) if re.search(r'adamw', r[2])]
    phases = {'forward + early backward': defaultdict(float), 'final backward': defaultdict(float)}
    spans = {'forward + early backward': 0.0, 'final backward': 0.0}
    n_ph = 0
    for mi in marks[-(steps + 1):-1]:
        prev_opt = max([o for o in opt if o < mi], default=None)
        next_opt = min([o for o in opt if o > mi], default=None)
        if prev_opt is None or next_opt is None:
            continue
        n_ph += 1
        for name, lo_i, hi_i in (('forward + early backward', prev_opt, mi), ('final backward', mi, next_opt)):
            spans[name] += (rows[hi_i][0] - rows[lo_i][0]) / 1e6
            for s, e, n, q, st in rows[lo_i:hi_i]:
                for cname, pat in CATS:
                    if re.search(pat, n):
                        break
                else:
                    cname = 'other own kernels'
                phases[name][cname] += (e - s) / 1e6
    if n_ph:
        for name, d in phases.items():
            print('%s: %.1f ms of wall per step; kernel ms: %s' % (name, spans[name] / n_ph, ', '.join(
                '%s %.1f' % (k, v / n_ph) for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:9])))
    edges = [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 500), (500, 10 ** 9)]
    print('idle gaps of the union (us): ' + ', '.join(
        '%s-%s: %d x = %.2f ms' % (lo_, hi_ if hi_ < 10 ** 9 else 'inf', sum(1 for g in gaps if lo_ * 1e3 <= g < hi_ * 1e3) / steps,
                                   sum(g for g in gaps if lo_ * 1e3 <= g < hi_ * 1e3) / 1e6 / steps) for lo_, hi_ in edges))


if __name__ == '__main__':
    main()
