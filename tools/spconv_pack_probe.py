"""Experiment: does grouping output rows with equal / similar neighbour masks into the same 16-row tile
speed up the gather-GEMM?  For every layer: tile fill and kernel time with rows in rulebook order vs
rows sorted by (a) the 27-bit mask, (b) mask with frequency-ordered bits.  Timing only: the outputs come
out in the permuted order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from detmatch_amd import _lib, synth, voxel  # noqa: E402
from detmatch_amd.pcdet.workload import BACKBONE_LAYERS  # noqa: E402
from detmatch_amd.spconv import ops  # noqa: E402
from bench_spconv_layers import timed  # noqa: E402


def fill(nbr):
    kvol, n = nbr.shape
    t = (n + 15) // 16
    pad = torch.full((kvol, t * 16), -1, dtype=nbr.dtype, device=nbr.device)
    pad[:, :n] = nbr
    v = (pad >= 0).view(kvol, t, 16)
    active = v.any(dim=2).sum().item()
    return v.sum().item() / max(active * 16, 1), active


def main():
    dev = torch.device('cuda:0')
    pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
    _, coors, _, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    idx, shape = coors, [41, 1600, 1408]
    books = {}
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key not in books:
            books[key] = ops.build_rulebook(idx, 2, shape, ks, st, pd, 1, subm)
        rb = books[key]
        if cin >= 32:
            x = torch.randn(rb.n_in, cin, device=dev)
            w = torch.randn(*ks, cin, cout, device=dev) * 0.05
            nbr = rb.nbr_out
            kvol = nbr.shape[0]
            valid = nbr >= 0
            weights = (1 << torch.arange(kvol, device=dev, dtype=torch.int64))[:, None]
            mask = (valid.long() * weights).sum(0)
            freq = valid.sum(1)
            order_bits = torch.argsort(freq, descending=True)        # most frequent offset = highest bit
            w2 = torch.zeros(kvol, dtype=torch.int64, device=dev)
            w2[order_bits] = 1 << torch.arange(kvol - 1, -1, -1, device=dev, dtype=torch.int64)
            mask2 = (valid.long() * w2[:, None]).sum(0)
            res = []
            for name, perm in (('rulebook order', None), ('sorted by mask', torch.sort(mask, stable=True)[1]),
                               ('sorted, frequent offsets first', torch.sort(mask2, stable=True)[1])):
                t = nbr if perm is None else nbr[:, perm].contiguous()
                f, active = fill(t)
                us = timed(lambda: ops._gather_gemm(x, w, t, rb.n_out, cin, cout, 0, 0), 30, 0)
                res.append('%s: fill %.2f units %d  %.1f us' % (name, f, active, us))
            print('%-8s %d>%d rows %d | %s' % (key, cin, cout, rb.n_out, ' | '.join(res)))
        idx, shape = rb.outids, rb.out_shape


if __name__ == '__main__':
    main()
