"""Per-layer timing of the hand-written dense convolutions at the REAL shapes of the step
(B = 2): forward / input gradient / weight gradient, TFLOP/s vs the 157.3 TFLOP/s fp32 matrix peak,
with torch's MIOpen convolution on the same shapes beside it.
    python tools/bench_dense_conv.py [--no-miopen]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
import torch.nn.functional as F
from detmatch_amd import dense_conv

PEAK = 157.3
LAYERS = [
    # name, (N,Cin,H,W), Cout, k, s, p
    ('bev b1.0 256->128', (2, 256, 200, 176), 128, 3, 1, 1),
    ('bev b1.x 128->128', (2, 128, 200, 176), 128, 3, 1, 1),
    ('bev b2.0 128->256 s2', (2, 128, 200, 176), 256, 3, 2, 1),
    ('bev b2.x 256->256', (2, 256, 100, 88), 256, 3, 1, 1),
    ('heads 1x1 512->72', (2, 512, 200, 176), 72, 1, 1, 0),
    ('stem 7x7 3->64 s2', (2, 3, 384, 1280), 64, 7, 2, 3),
    ('r50 l1 1x1 64->64', (2, 64, 96, 320), 64, 1, 1, 0),
    ('r50 l1 3x3 64->64', (2, 64, 96, 320), 64, 3, 1, 1),
    ('r50 l1 1x1 64->256', (2, 64, 96, 320), 256, 1, 1, 0),
    ('r50 l2 3x3 128->128', (2, 128, 48, 160), 128, 3, 1, 1),
    ('r50 l2 1x1 512->128', (2, 512, 48, 160), 128, 1, 1, 0),
    ('r50 l3 3x3 256->256', (2, 256, 24, 80), 256, 3, 1, 1),
    ('r50 l3 1x1 1024->256', (2, 1024, 24, 80), 256, 1, 1, 0),
    ('r50 l4 3x3 512->512', (2, 512, 12, 40), 512, 3, 1, 1),
    ('fpn 3x3 256->256 P2', (2, 256, 96, 320), 256, 3, 1, 1),
    ('fpn 3x3 256->256 P4', (2, 256, 24, 80), 256, 3, 1, 1),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device('cuda:0')
    miopen = '--no-miopen' not in sys.argv
    print('%-26s %9s | %8s %6s | %8s %6s | %8s %6s || MIOpen fwd / dgrad / wgrad (us)' %
          ('layer', 'GFLOP', 'fwd us', 'TF/s', 'dgrad us', 'TF/s', 'wgrad us', 'TF/s'))
    tot = [0.0] * 6
    for name, xs, cout, k, s, p in LAYERS:
        x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(cout, xs[1], k, k, device=dev) * 0.05)
        y = dense_conv.conv2d(x, w, None, s, p)
        gf = 2.0 * y.numel() * xs[1] * k * k / 1e9
        dy = torch.randn_like(y)
        xg = x.clone().requires_grad_(xs[1] >= 4)

        def fwd():
            return dense_conv.conv2d(x, w, None, s, p)

        def bwd(which):
            yy = dense_conv.conv2d(xg if which != 'w' else x, w if which != 'x' else w.detach(), None, s, p)
            return yy

        t_f = timeit(fwd)
        # input gradient only / weight gradient only (forward time subtracted)
        def dgrad():
            yy = dense_conv.conv2d(xg, w.detach(), None, s, p)
            yy.backward(dy)
        def wgrad():
            yy = dense_conv.conv2d(x, w, None, s, p)
            yy.backward(dy)
        t_d = timeit(dgrad) - t_f if xs[1] >= 4 else float('nan')
        t_w = timeit(wgrad) - t_f
        row = '%-26s %9.2f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f' % (
            name, gf, t_f * 1e6, gf / t_f / 1e3, t_d * 1e6, gf / t_d / 1e3, t_w * 1e6, gf / t_w / 1e3)
        if miopen:
            wm = w.detach().clone().requires_grad_(True)
            xm = x.clone().requires_grad_(xs[1] >= 4)
            m_f = timeit(lambda: F.conv2d(x, wm.detach(), None, s, p))
            def md():
                F.conv2d(xm, wm.detach(), None, s, p).backward(dy)
            def mw():
                F.conv2d(x, wm, None, s, p).backward(dy)
            m_d = timeit(md) - m_f if xs[1] >= 4 else float('nan')
            m_w = timeit(mw) - m_f
            row += ' || %8.1f %8.1f %8.1f' % (m_f * 1e6, m_d * 1e6, m_w * 1e6)
        print(row, flush=True)
    print('peak fp32 matrix: %.1f TFLOP/s' % PEAK)


if __name__ == '__main__':
    main()
