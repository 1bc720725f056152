"""Shared-MLP GEMMs of the set abstraction / RoI-grid pooling (rows x K -> N, rows in the hundreds of
thousands): torch's BLAS call against the 1x1 path of the hand-written convolution kernel on the same
row-major matrix, forward and input gradient; HBM floor = rows * (K + N) * 4 bytes at 7.3 TB/s.
    python tools/bench_tall_skinny.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detmatch_amd import dense_conv  # noqa: E402


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda', 0)
    shapes = [(884736, 132, 64), (884736, 64, 64), (131072, 68, 64), (131072, 64, 64), (65536, 36, 32),
              (65536, 32, 32), (131072, 20, 16), (65536, 16, 16)]
    print('%-22s | %9s %9s | %9s %9s | %7s' % ('rows x K -> N', 'mm fwd', 'conv fwd', 'mm dgrad', 'conv dgr', 'floor'))
    for r, k, n in shapes:
        x = torch.randn(r, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        gy = torch.randn(r, n, device=dev)
        mm_f = t(lambda: x @ w.t())
        mm_d = t(lambda: gy @ w)
        x4 = x.view(1, r, 1, k).permute(0, 3, 1, 2)
        w4 = torch.nn.Parameter(w.view(n, k, 1, 1).contiguous())
        with torch.no_grad():
            cv_f = t(lambda: dense_conv.conv2d(x4, w4))
        xg = x4.detach().requires_grad_(True)
        y = dense_conv.conv2d(xg, w4)
        g4 = gy.view(1, r, 1, n).permute(0, 3, 1, 2)
        w4.requires_grad_(False)
        cv_d = t(lambda: torch.autograd.grad(y, xg, g4, retain_graph=True))
        from detmatch_amd.pointnet2_stack import TallSkinnyLinear
        TallSkinnyLinear.ROWGEMM_MIN_ROWS = 0
        wt = w.t().contiguous()
        rg_f = t(lambda: TallSkinnyLinear._rowgemm(x, w))
        rg_d = t(lambda: TallSkinnyLinear._rowgemm(gy, wt))
        err = float((TallSkinnyLinear._rowgemm(x, w) - x @ w.t()).abs().max())
        floor = r * (k + n) * 4 / 7.3e12 * 1e6
        print('%8d x %3d -> %3d | %9.1f %9.1f | %9.1f %9.1f | %7.1f | rowgemm fwd %7.1f dgrad %7.1f  err %.1e' % (r, k, n, mm_f, cv_f, mm_d, cv_d, floor, rg_f, rg_d, err))


if __name__ == '__main__':
    main()
