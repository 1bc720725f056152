"""Forward time of the dense-conv layers in both arithmetic modes (fp32 exact / bf16 multiplicands):
kernel time from HIP events around 10 back-to-back launches, best of four windows.
    python tools/bench_dense_conv_math.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detmatch_amd import dense_conv
from bench_dense_conv import LAYERS


def main():
    dev = torch.device('cuda:0')
    print('%-26s %8s | %9s %7s | %9s %7s %5s | %9s %7s %5s | %9s' % ('layer', 'GFLOP', 'fp32 MFMA', 'TF/s', 'split us', 'TF/s',
                                                                     'x', 'bf16 us', 'TF/s', 'x', 'no patch'))
    for name, xs, cout, k, s, p in LAYERS:
        if xs[1] % 32:
            continue
        x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(cout, xs[1], k, k, device=dev) * 0.05)
        ho, wo = (xs[2] + 2 * p - k) // s + 1, (xs[3] + 2 * p - k) // s + 1
        gf = 2.0 * xs[0] * ho * wo * cout * xs[1] * k * k / 1e9
        res = []
        for mode in ('fp32_mfma', 'fp32_split', 'bf16', 'fp32_split_nopatch'):
            dense_conv.set_math(mode)
            with torch.no_grad():
                for _ in range(3):
                    dense_conv.conv2d(x, w, None, s, p)
                best = float('inf')
                for _ in range(4):      # best of four windows of ten launches: one allocator stall does not land in a cell
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(10):
                        dense_conv.conv2d(x, w, None, s, p)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            res.append(best)
        dense_conv.set_math('fp32')
        print('%-26s %8.2f | %9.1f %7.1f | %9.1f %7.1f %5.2f | %9.1f %7.1f %5.2f | %9.1f' % (
            name, gf, res[0], gf / res[0] * 1e3, res[1], gf / res[1] * 1e3, res[0] / res[1], res[2], gf / res[2] * 1e3,
            res[0] / res[2], res[3]))


if __name__ == '__main__':
    main()
