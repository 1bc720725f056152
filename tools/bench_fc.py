"""Fully connected layers of the two RoI heads: torch (hipBLASLt / Tensile) against the own lattice GEMM
(csrc/conv2d.hip through dense_conv.conv2d on (M, K, 1, 1) views), forward / input gradient / weight gradient.

    python tools/bench_fc.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from detmatch_amd import _lib, dense_conv  # noqa: E402

SHAPES = [('pvrcnn shared_fc 27648->256, 256 rois', 256, 27648, 256),
          ('pvrcnn shared_fc, teacher 200 rois', 200, 27648, 256),
          ('pvrcnn 256->256', 256, 256, 256),
          ('frcnn fc1 12544->1024, 1024 rois', 1024, 12544, 1024),
          ('frcnn fc1, teacher 2000 rois', 2000, 12544, 1024),
          ('frcnn fc2 1024->1024', 1024, 1024, 1024),
          ('frcnn cls+reg 1024->16', 1024, 1024, 16),
          ('vsa fusion 640->128, 4096 pts', 4096, 640, 128),
          ('point head 128->256', 4096, 128, 256)]


def timed(fn, reps=20):
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


def main():
    dev = torch.device('cuda', 0)
    print('%-40s %6s | %30s | %30s | %30s' % ('layer', 'GFLOP', 'fwd  blas / conv / fc_gemm  us', 'dgrad blas / conv / fc_gemm', 'wgrad blas / conv / fc_gemm'))
    for name, m, k, n in SHAPES:
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        gy = torch.randn(m, n, device=dev)
        xc, wc = x.view(m, k, 1, 1), w.view(n, k, 1, 1)
        y_b = F.linear(x, w)
        y_o = dense_conv.conv2d(xc, wc).view(m, n)
        err = float((y_b - y_o).abs().max() / y_b.abs().max())
        t_fb = timed(lambda: F.linear(x, w))
        t_fo = timed(lambda: dense_conv.conv2d(xc, wc))
        t_db = timed(lambda: gy @ w)
        t_wb = timed(lambda: gy.t() @ x)
        xo = xc.clone().requires_grad_(True)
        wo = wc.clone().requires_grad_(True)

        def own_bwd(need_x, need_w):
            xo.requires_grad_(need_x)
            wo.requires_grad_(need_w)
            y = dense_conv.conv2d(xo, wo)
            y.backward(gy.view(m, n, 1, 1))
            xo.grad = wo.grad = None
        t_f2 = timed(lambda: dense_conv.conv2d(xo.detach(), wo.detach()))
        t_do = timed(lambda: own_bwd(True, False)) - t_f2
        t_wo = timed(lambda: own_bwd(False, True)) - t_f2
        gf = 2.0 * m * k * n / 1e9
        yb, gxb, gwb = torch.empty(m, n, device=dev), torch.empty(m, k, device=dev), torch.empty(n, k, device=dev)
        t_ff = timed(lambda: _lib._fc_gemm(0, x, w, None, yb, m, n, k, k, k))
        t_df = timed(lambda: _lib._fc_gemm(1, gy, w, None, gxb, m, k, n, n, k))
        t_wf = timed(lambda: _lib._fc_gemm(2, gy, x, None, gwb, n, k, m, n, k))
        err_f = float((yb - y_b).abs().max() / y_b.abs().max())
        print('%-40s %6.2f | %8.1f / %8.1f / %8.1f | %8.1f / %8.1f / %8.1f | %8.1f / %8.1f / %8.1f   err conv %.1e fc %.1e' % (
            name, gf, t_fb, t_fo, t_ff, t_db, t_do, t_df, t_wb, t_wo, t_wf, err, err_f))


if __name__ == '__main__':
    main()
