"""the REFERENCE's own CPU kernels compiled here (oracle/_ref: voxelizer, geometry.h +
reordering.cc with torch::mm_out between gather and scatter, iou3d_cpu.cpp) timed beside the C port
(oracle/dm_oracle.c) on the bench frames — KITTI-shaped synthetic frames 0 and 1, B = 2, the 12 layers / 8
rulebooks of VoxelBackBone8x — at 1 and 8 threads.  Build container only (needs /root/reference).

    python tools/cpu_baseline_table.py > profiles/r03_cpu_reference_vs_port.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from detmatch_amd import synth  # noqa: E402
from oracle import build_ref  # noqa: E402
from gen_spconv_golden import LAYERS, SHAPE, layer_weight, out_size  # noqa: E402


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e3, r


def main():
    oracle.build()
    build_ref.build_all()
    ref = build_ref.load_ref(build_ref.SPCONV_NAME)
    iou_ref = build_ref.load_ref(build_ref.IOU3D_NAME)
    frames = [synth.lidar_frame(s) for s in (0, 1)]
    print('host: %d cores; torch %s; frames: %s points' % (os.cpu_count(), torch.__version__,
                                                            [len(f['points']) for f in frames]))
    rows = []
    for threads in (1, 8):
        torch.set_num_threads(threads)
        # ---- voxelize
        def vox_ref():
            return [build_ref.ref_hard_voxelize(f['points'], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000) for f in frames]

        def vox_port():
            return [oracle.hard_voxelize(f['points'], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000) for f in frames]
        vox_ref()          # first touch of the reference's 360 MB grid
        t_vr, vr = best(vox_ref)
        t_vp, vp = best(vox_port)
        feats = np.concatenate([(v.sum(1) / np.maximum(n, 1)[:, None]).astype(np.float32) for v, c, n in vp])
        idx = np.concatenate([np.concatenate([np.full((len(n), 1), b, np.int32), c], 1) for b, (v, c, n) in enumerate(vp)]).astype(np.int32)

        # ---- rulebooks
        def rb_ref():
            books, shape, cur = {}, SHAPE, torch.from_numpy(idx)
            for key, subm, cin, cout, ks, st, pd in LAYERS:
                if key in books:
                    continue
                osh = shape if subm else out_size(shape, ks, st, pd)
                o, p, n = ref.get_indice_pairs(cur, 2, osh, ks, st, pd, [1, 1, 1], subm)
                books[key] = (o, p, n)
                cur, shape = o, osh
            return books

        def rb_port():
            books, shape, cur = {}, SHAPE, idx
            for key, subm, cin, cout, ks, st, pd in LAYERS:
                if key in books:
                    continue
                o, p, n, osh = oracle.get_indice_pairs(cur, 2, shape, ks, st, pd, subm=subm)
                books[key] = (o, p, n)
                cur, shape = o, osh
            return books
        t_rr, books_r = best(rb_ref, 2)
        t_rp, books_p = best(rb_port, 2)
        ws = [layer_weight(li, ks, cin, cout) for li, (_, _, cin, cout, ks, _, _) in enumerate(LAYERS)]

        # ---- 12 convs forward + backward
        def conv_ref():
            x = torch.from_numpy(feats)
            acts = []
            t0 = time.perf_counter()
            for (key, subm, cin, cout, ks, st, pd), w in zip(LAYERS, ws):
                o, p, n = books_r[key]
                y = ref.indice_conv(x, torch.from_numpy(w), p, n, o.shape[0], False, subm)
                acts.append((x, torch.from_numpy(w), p, n, subm))
                x = torch.relu(y)
            tf = time.perf_counter() - t0
            g = torch.ones_like(x)
            t0 = time.perf_counter()
            for xi, w, p, n, subm in reversed(acts):
                g, _ = ref.indice_conv_backward(xi, w, g, p, n, False, subm)
            return tf * 1e3, (time.perf_counter() - t0) * 1e3

        def conv_port():
            x = feats
            acts = []
            t0 = time.perf_counter()
            for (key, subm, cin, cout, ks, st, pd), w in zip(LAYERS, ws):
                o, p, n = books_p[key]
                y = oracle.indice_conv(x, w.reshape(-1, cin, cout), p, n, len(o), subm=subm)
                acts.append((x, w.reshape(-1, cin, cout), p, n, subm))
                x = np.maximum(y, 0)
            tf = time.perf_counter() - t0
            g = np.ones_like(x)
            t0 = time.perf_counter()
            for xi, w, p, n, subm in reversed(acts):
                g, _ = oracle.indice_conv_backward(xi, w, g, p, n, subm=subm)
            return tf * 1e3, (time.perf_counter() - t0) * 1e3
        cr = min((conv_ref() for _ in range(2)), key=sum)
        cp = min((conv_port() for _ in range(2)), key=sum) if threads == 1 else None
        # ---- BEV IoU
        rng = np.random.default_rng(1)
        nb = 256
        boxes = np.concatenate([rng.uniform(0, 40, (nb, 2)), rng.uniform(-1, 1, (nb, 1)), rng.uniform(1, 4, (nb, 3)),
                                rng.uniform(-3, 3, (nb, 1))], 1).astype(np.float32)
        tb = torch.from_numpy(boxes)
        ans = torch.zeros(nb, nb)
        t_ir, _ = best(lambda: iou_ref.boxes_iou_bev_cpu(tb, tb, ans))
        t_ip, _ = best(lambda: oracle.boxes_iou_bev(boxes, boxes))
        rows.append((threads, t_vr / 2, t_vp / 2, t_rr, t_rp, cr, cp, t_ir * 1e3 / nb / nb, t_ip * 1e3 / nb / nb))
    print()
    print('%-34s %14s %14s %14s' % ('piece', 'reference, 1 thr', 'reference, 8 thr', 'C port, 1 thr'))
    r1, r8 = rows
    fmt = '%-34s %14.1f %14.1f %14s'
    print(fmt % ('hard voxelize, ms / frame', r1[1], r8[1], '%.1f' % r1[2]))
    print(fmt % ('8 rulebooks, ms', r1[3], r8[3], '%.1f' % r1[4]))
    print(fmt % ('12 sparse convs forward, ms', r1[5][0], r8[5][0], '%.1f' % r1[6][0]))
    print(fmt % ('... backward (dgrad + wgrad), ms', r1[5][1], r8[5][1], '%.1f' % r1[6][1]))
    print('%-34s %14.3f %14.3f %14s' % ('rotated BEV IoU, us / pair', r1[7], r8[7], '%.3f' % r1[8]))
    print()
    print('port / reference (1 thread): voxelize %.2fx, rulebooks %.2fx, conv fwd %.2fx, conv bwd %.2fx, IoU %.2fx'
          % (r1[2] / r1[1], r1[4] / r1[3], r1[6][0] / r1[5][0], r1[6][1] / r1[5][1], r1[8] / r1[7]))
    print('(the port is plain C, one thread, its GEMM loops auto-vectorised by gcc -O3 -mavx2 without reassociation; '
          'the reference path runs its per-offset GEMMs through MKL — torch::mm_out — which is why its convolutions '
          'are faster, and allocates a dense 360 MB grid per voxelizer call, which is why its voxelizer is slower)')


if __name__ == '__main__':
    main()
