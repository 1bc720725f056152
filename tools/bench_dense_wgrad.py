"""Weight-gradient time of the 3 x 3 / stride-1 dense-conv layers in the split arithmetic: the tap-fused kernel
(dconv_wgrad9_kernel, default) against one workgroup set per tap ('fp32_split_tapwise_wgrad').  Kernel + reduce time
from HIP events around 10 back-to-back calls, best of four windows.
    python tools/bench_dense_wgrad.py [substring]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detmatch_amd import dense_conv
from bench_dense_conv import LAYERS


def main():
    dev = torch.device('cuda:0')
    want = sys.argv[1] if len(sys.argv) > 1 else ''
    print('%-26s %8s | %10s %7s | %10s %7s | %5s' % ('layer', 'GFLOP', 'per-tap us', 'TF/s', 'fused us', 'TF/s', 'x'))
    tot = [0.0, 0.0]
    for name, xs, cout, k, s, p in LAYERS:
        if xs[1] % 4 or k != 3 or s != 1 or want not in name:
            continue
        n, cin, h, w = xs
        x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn((n, cout, h, w), device=dev).contiguous(memory_format=torch.channels_last)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        taps = [(a - 1, b - 1) for a in range(3) for b in range(3)]
        geom = [n, h, w, cout, cin, h, w, 1, 1, 9]
        gf = 2.0 * n * h * w * cout * cin * 9 / 1e9
        res, outs = [], []
        for mode in ('fp32_split_tapwise_wgrad', 'fp32_split'):
            dense_conv.set_math(mode)
            for _ in range(3):
                dense_conv._wgrad(dy, x, dw, None, geom, taps, cin, cin * 9, 9, 1)
            best = float('inf')
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    dense_conv._wgrad(dy, x, dw, None, geom, taps, cin, cin * 9, 9, 1)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            res.append(best)
            outs.append(dw.clone())
        dense_conv.set_math('fp32')
        d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
        tot[0] += res[0]
        tot[1] += res[1]
        print('%-26s %8.2f | %10.1f %7.1f | %10.1f %7.1f | %5.2f   rel diff %.1e' % (
            name, gf, res[0], gf / res[0] * 1e3, res[1], gf / res[1] * 1e3, res[0] / res[1], d))
    print('sum: per-tap %.1f us, fused %.1f us' % tuple(tot))


if __name__ == '__main__':
    main()
