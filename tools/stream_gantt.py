"""Per-stream picture of ONE steady-state DetMatch iteration from a rocprofv3 --kernel-trace CSV: for every HIP stream the
busy intervals (kernels closer than `--gap` us merged) with their dominant kernel families, on a common time axis that
starts at the iteration's first kernel.  Shows which lane the device is waiting for where.

    python tools/stream_gantt.py <..._kernel_trace.csv> [--marker ema_f32] [--skip-last 3] [--gap 150]
"""
import argparse
import csv
import re
from collections import Counter, defaultdict

FAM = [('dconv', r'dconv_'), ('bn', r'bn_'), ('spconv', r'spconv|rb_|tile_order|pack_rows|voxel'), ('rowgemm', r'rowgemm'),
       ('fc', r'fc_gemm|fc_reduce'), ('sa', r'ball_query|group_|query_group|tall_wgrad'), ('fps', r'fps_'),
       ('roialign', r'roi_align'), ('nms', r'nms_|iou'), ('sort', r'sort|rocprim'), ('aten', r'at::native|rocclr'),
       ('opt', r'adamw|sgd_|ema_')]


def fam(name):
    for f, pat in FAM:
        if re.search(pat, name):
            return f
    return 'own'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--marker', default='ema_f32')
    ap.add_argument('--skip-last', type=int, default=0)
    ap.add_argument('--pick', type=int, default=-1)
    ap.add_argument('--gap', type=float, default=150.0)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '?')))
    rows.sort()
    opt = [i for i, r in enumerate(rows) if re.search(r'adamw', r[2])]
    # the timed region = the run of back-to-back iterations: take the interval of median length among the shortest half
    gaps = sorted((rows[opt[i + 1]][0] - rows[opt[i]][0], i) for i in range(len(opt) - 1))
    short = gaps[:max(1, len(gaps) // 2)]
    pick = short[len(short) // 2][1] if a.pick < 0 else a.pick
    print('optimizer-to-optimizer intervals (ms): ' + ' '.join('%.1f' % ((rows[opt[i + 1]][0] - rows[opt[i]][0]) / 1e6)
                                                                for i in range(len(opt) - 1)) + '   -> picked #%d' % pick)
    lo, hi = opt[pick] + 1, opt[pick + 1] + 1    # one iteration: behind the previous optimizer step .. this optimizer step
    # (the SGD step follows AdamW: extend to the next kernel that is not an optimizer kernel)
    while hi < len(rows) and re.search(r'sgd_|adamw', rows[hi][2]):
        hi += 1
    win = rows[lo:hi]
    t0 = win[0][0]
    print('iteration window: %.2f ms, %d kernels' % ((win[-1][1] - t0) / 1e6, len(win)))
    ema = [r for r in win if re.search(a.marker, r[2])]
    if ema:
        print('EMA (end of forward_train) at %.2f ms' % ((ema[0][0] - t0) / 1e6))
    per = defaultdict(list)
    for s, e, n, st in win:
        per[st].append((s, e, n))
    for st in sorted(per, key=lambda k: -sum(e - s for s, e, _ in per[k])):
        ks = per[st]
        busy = sum(e - s for s, e, _ in ks) / 1e6
        print('\nstream %s: %d kernels, busy %.2f ms' % (st, len(ks), busy))
        cur_s, cur_e, names, t = ks[0][0], ks[0][1], Counter(), 0.0
        names[fam(ks[0][2])] += ks[0][1] - ks[0][0]
        out = []
        for s, e, n in ks[1:]:
            if s - cur_e > a.gap * 1e3:
                out.append((cur_s, cur_e, names))
                cur_s, cur_e, names = s, e, Counter()
            else:
                cur_e = max(cur_e, e)
            names[fam(n)] += e - s
        out.append((cur_s, cur_e, names))
        for s, e, names in out:
            if (e - s) < 100e3 and len(out) > 40:
                continue
            top = ', '.join('%s %.1f' % (k, v / 1e6) for k, v in names.most_common(4))
            print('   %7.2f .. %7.2f  (%5.2f ms, kernels %.2f ms)  %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6,
                                                                        sum(names.values()) / 1e6, top))


if __name__ == '__main__':
    main()
