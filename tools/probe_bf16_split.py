"""How accurate is an fp32 convolution assembled from bf16 x bf16 products (fp32 accumulate) of the 3-way
bf16 split of both operands?  x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) carries the
24 significand bits of an fp32 number; the 9 cross products of two such splits are exact in fp32.  The
existing mixed-precision kernel (csrc/conv2d.hip, v_mfma_f32_32x32x16_bf16) is run on pre-split operands
(rounding a bf16-representable value is the identity) and the partial outputs are summed in fp32, smallest
terms first.  Reference: float64 convolution on the CPU.

    python tools/probe_bf16_split.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from detmatch_amd import dense_conv  # noqa: E402


def split3(x):
    h = x.bfloat16().float()
    r = x - h
    m = r.bfloat16().float()
    l = (r - m).bfloat16().float()
    return h, m, l


def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for name, (n, c, hh, ww, co, k) in {'bev 3x3 128->128': (2, 128, 100, 88, 128, 3),
                                        'r50 1x1 512->128': (2, 512, 48, 160, 128, 1),
                                        'positive data (ReLU outputs, biased sums)': (2, 128, 50, 44, 128, 3)}.items():
        x = torch.randn(n, c, hh, ww, generator=g)
        w = torch.randn(co, c, k, k, generator=g) / (c * k * k) ** 0.5
        if 'positive' in name:
            x, w = x.abs(), w.abs()
        y64 = F.conv2d(x.double(), w.double(), None, 1, k // 2)
        scale = float(y64.abs().max())
        xd, wd = x.to(dev), w.to(dev)
        dense_conv.set_math('fp32')
        y32 = dense_conv.conv2d(xd, wd, None, 1, k // 2).cpu().double()
        xs, ws = split3(xd), split3(wd)
        assert float((xs[0] + xs[1] + xs[2] - xd).abs().max()) == 0.0 or True
        rec = float((xs[0].double() + xs[1].double() + xs[2].double() - xd.double()).abs().max() / xd.abs().max())
        dense_conv.set_math('bf16')
        part = {}
        for i in range(3):
            for j in range(3):
                part[(i, j)] = dense_conv.conv2d(xs[i], ws[j], None, 1, k // 2)
        dense_conv.set_math('fp32')
        order = sorted(part, key=lambda ij: -(ij[0] + ij[1]))          # smallest terms first

        def total(keep):
            acc = torch.zeros_like(part[(0, 0)])
            for ij in order:
                if keep(ij):
                    acc = acc + part[ij]
            return acc.cpu().double()
        res = {'native fp32 MFMA': y32, 'bf16 x9': total(lambda ij: True),
               'bf16 x6 (i + j <= 2)': total(lambda ij: ij[0] + ij[1] <= 2),
               'bf16 x3 (i + j <= 1)': total(lambda ij: ij[0] + ij[1] <= 1), 'bf16 x1': total(lambda ij: ij == (0, 0))}
        print('%s   (split reconstruction error %.1e)' % (name, rec))
        for kname, y in res.items():
            e = (y - y64).abs()
            print('   %-24s max err / max|y| = %.2e   rms err / rms y = %.2e' % (
                kname, float(e.max()) / scale, float(e.pow(2).mean().sqrt() / y64.pow(2).mean().sqrt())))


if __name__ == '__main__':
    main()
