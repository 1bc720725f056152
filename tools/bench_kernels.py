"""Per-kernel timings of the C-ABI operators on the bench shapes (KITTI-shaped synthetic, B=2):
us per call and algorithmic GB/s (SURVEY §8d byte formulas).  GPU only.

    python tools/bench_kernels.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from detmatch_amd import _lib, iou3d_nms, pointnet2_stack as pn, synth, voxel  # noqa: E402
from detmatch_amd.roi_align import roi_align_fpn  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def row(name, us, nbytes=None, note=''):
    gbs = ('%8.0f' % (nbytes / us / 1e3)) if nbytes else '       -'
    print('%-44s %9.1f us %s GB/s  %s' % (name, us, gbs, note))


def main():
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    frames = [synth.lidar_frame(s) for s in range(2)]
    pts = [torch.from_numpy(f['points']).to(dev) for f in frames]
    n_p = sum(len(p) for p in pts)
    v, c, n, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000, with_mean=True)
    V = v.shape[0]
    us = timed(lambda: voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000, with_mean=True))
    row('hard_voxelize + MeanVFE (B=2, incl. count read-back)', us, n_p * 16 + V * (5 * 16 + 12 + 4),
        '%d pts -> %d voxels' % (n_p, V))
    # ---- FPS / ball query / grouping -------------------------------------------------------
    xyz = torch.cat([p[:, :3] for p in pts]).contiguous()
    cnt = [len(p) for p in pts]
    us = timed(lambda: pn.furthest_point_sample_stack(xyz, cnt, 2048), 5)
    row('furthest_point_sample_stack (2 x ~20k -> 2048)', us, n_p * 12, 'latency bound: 2047 dependent rounds')
    kp_idx = pn.furthest_point_sample_stack(xyz, cnt, 2048).long()
    off = 0
    kps = []
    for b, k in enumerate(cnt):
        kps.append(xyz[off:off + k][kp_idx[b]])
        off += k
    kp = torch.cat(kps).contiguous()
    tcnt = torch.tensor(cnt, dtype=torch.int32, device=dev)
    kcnt = torch.tensor([2048, 2048], dtype=torch.int32, device=dev)
    us = timed(lambda: pn.ball_query(0.4, 16, xyz, tcnt, kp, kcnt))
    row('ball_query raw points r=0.4 ns=16 (M=4096, N=%d)' % n_p, us, n_p * 12 + 4096 * 12 + 4096 * 16 * 4)
    # RoI-grid shapes: 55296 queries over 4096 key-points, C=128
    rois_c = kp[rng.integers(0, 4096, 256)]
    grid = (rois_c[:, None, :] + torch.from_numpy(rng.uniform(-1.5, 1.5, (256, 216, 3)).astype(np.float32)).to(dev))
    grid = grid.view(-1, 3).contiguous()
    # queries must be grouped per sample: sort by sample of their RoI
    gcnt = torch.tensor([27648, 27648], dtype=torch.int32, device=dev)
    feats = torch.randn(4096, 128, device=dev, requires_grad=True)
    us = timed(lambda: pn.ball_query(0.8, 16, kp, kcnt, grid, gcnt))
    row('ball_query RoI grid r=0.8 ns=16 (M=55296, N=4096)', us, 4096 * 12 + 55296 * 12 + 55296 * 16 * 4)
    idx, empty = pn.ball_query(0.8, 16, kp, kcnt, grid, gcnt)
    us = timed(lambda: pn.QueryGroupRows.apply(kp, kcnt, grid, gcnt, feats, idx, empty, True))
    row('query_group_rows (M=55296, ns=16, C=3+128)', us, 55296 * 16 * 131 * 4 * 2)
    rows = pn.QueryGroupRows.apply(kp, kcnt, grid, gcnt, feats, idx, empty, True)
    g = torch.randn_like(rows)
    us = timed(lambda: torch.autograd.grad(rows, feats, g, retain_graph=True))
    row('group_rows_grad (LDS-combining)', us, 55296 * 16 * 131 * 4 + 4096 * 128 * 4)
    # ---- rotated NMS -------------------------------------------------------------------------
    b = np.concatenate([rng.uniform(0, 70, (9000, 1)), rng.uniform(-40, 40, (9000, 1)), rng.uniform(-1, 1, (9000, 1)),
                        rng.uniform(1.5, 4.5, (9000, 3)), rng.uniform(-3.2, 3.2, (9000, 1))], 1).astype(np.float32)
    tb = torch.from_numpy(b).to(dev)
    ts = torch.from_numpy(rng.permutation(9000).astype(np.float32)).to(dev)
    us = timed(lambda: iou3d_nms.nms_gpu(tb, ts, 0.8, post_max_size=512), 10)
    row('rotated NMS 9000 -> 512 (two-phase mask + device greedy)', us, 9000 * 28 + 9000 * 141 * 8, 'incl. sort + count read-back')
    a, bb = tb[:512], tb[512:540]
    us = timed(lambda: iou3d_nms.boxes_iou3d_gpu(a, bb))
    row('boxes_iou3d 512 x 28', us, (512 + 28) * 28 + 512 * 28 * 4)
    # ---- RoIAlign over the pyramid -------------------------------------------------------------
    fpn = [torch.randn(2, 256, 96 // s, 320 // s, device=dev, requires_grad=True) for s in (1, 2, 4, 8)]
    cc = rng.uniform([0, 0], [1280, 384], (1024, 2))
    ss = rng.uniform(8, 300, (1024, 2)) * rng.uniform(0.1, 1, (1024, 1))
    rb = np.concatenate([cc - ss / 2, cc + ss / 2], 1)
    rb[:, 0::2] = rb[:, 0::2].clip(0, 1280)
    rb[:, 1::2] = rb[:, 1::2].clip(0, 384)
    rois = torch.from_numpy(np.concatenate([rng.integers(0, 2, (1024, 1)), rb], 1).astype(np.float32)).to(dev)
    us = timed(lambda: roi_align_fpn(fpn, rois, [4, 8, 16, 32]))
    row('roi_align_fpn forward (1024 RoIs, C=256, 4 levels)', us, 1024 * 256 * 49 * 4 * 2)
    out = roi_align_fpn(fpn, rois, [4, 8, 16, 32])
    go = torch.randn_like(out)
    us = timed(lambda: torch.autograd.grad(out, fpn, go, retain_graph=True), 10)
    row('roi_align_fpn backward (separable, NHWC atomics)', us, 1024 * 256 * 49 * 4 * 2, 'incl. zero-fill of the 4 grad maps')
    # ---- step driver ------------------------------------------------------------------------------
    n = 54_200_000
    p, g2, m, vv = (torch.randn(n, device=dev) for _ in range(4))
    vv.abs_()
    L = _lib.lib()
    us = timed(lambda: _lib.check(L.dm_adamw_step_f32(_lib.ptr(p), _lib.ptr(g2), _lib.ptr(m), _lib.ptr(vv), n, 1e-3, 0.95,
                                                      0.99, 1e-8, 0.01, 3, None, _lib.stream()), 'adamw'))
    row('fused AdamW step, 54.2 M params', us, n * 4 * 7)
    us = timed(lambda: _lib.check(L.dm_ema_update_f32(_lib.ptr(p), _lib.ptr(g2), n, 0.999, _lib.stream()), 'ema'))
    row('fused EMA, 54.2 M floats', us, n * 4 * 3)
    # ---- data pipeline: 8 views (4 frames x student/teacher) in one call -------------
    import ctypes
    from detmatch_amd import pipeline3d as P3
    for tag, reps_pts in (('KITTI', 1), ('Waymo-sized', 10)):
        raw = [torch.from_numpy(np.tile(synth.lidar_frame(s)['points'], (reps_pts, 1))).to(dev) for s in range(4)]
        views = []
        rs = np.random.RandomState(0)
        for i in range(4):
            sv, tv = P3.View3D(i), P3.View3D(i)
            P3.GlobalRotScaleTrans()(sv, rs)
            for v in (sv, tv):
                P3.PointsRangeFilter(synth.KITTI_RANGE)(v)
            views += [sv, tv]
        flat = torch.cat(raw)
        lens = [len(r) for r in raw]
        st = np.concatenate([[0], np.cumsum(lens)])
        src_off = [int(st[v.source]) for v in views]
        src_len = [lens[v.source] for v in views]
        dst_off = np.concatenate([[0], np.cumsum(src_len)])
        par = torch.from_numpy(np.stack([v.params() for v in views])).to(dev)
        out = torch.empty((int(dst_off[-1]), 4), device=dev)
        cnts = torch.empty(8, dtype=torch.int32, device=dev)
        ws = _lib.workspace(L.dm_points_augment_workspace_bytes(8, _lib.ints(src_len)), dev, 'aug')
        a = (_lib.ptr(flat), 4, 8, _lib.ints(src_off), _lib.ints(src_len), _lib.ints(dst_off[:-1]), _lib.ptr(par),
             None, _lib.ptr(out), _lib.ptr(cnts), _lib.ptr(ws), ctypes.c_size_t(ws.numel()))
        us = timed(lambda: _lib.check(L.dm_points_augment(*a, _lib.stream()), 'aug'))
        tot = int(dst_off[-1])
        row('dm_points_augment 8 views, %s (%d slots)' % (tag, tot), us, tot * 16 * 3,
            '2 launches; rows read twice (count + scatter), written once')


if __name__ == '__main__':
    main()
