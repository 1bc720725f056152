"""Furthest point sampling time: KITTI-sized (2 x 20 k -> 2048) and Waymo-sized (1 x 200 k -> 4096)
clouds, multi-workgroup kernel vs one workgroup per sample.
    python tools/bench_fps.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
from detmatch_amd import _lib, pointnet2_stack as pn2

dev = torch.device('cuda:0')
for n, m, b in ((20000, 2048, 2), (200000, 4096, 1), (200000, 4096, 2), (60000, 2048, 2)):
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand(n * b, 3, generator=g) * torch.tensor([150.0, 150.0, 6.0])).to(dev)
    res = []
    for v in (0, 1):
        _lib.lib().dm_fps_set_variant(v)
        pn2.furthest_point_sample_stack(xyz, [n] * b, m)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = pn2.furthest_point_sample_stack(xyz, [n] * b, m)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 3, out))
    _lib.lib().dm_fps_set_variant(0)
    print('%d x %6d -> %4d : auto %8.2f ms (%.2f us per round) | one workgroup per sample %8.2f ms | same indices: %s'
          % (b, n, m, res[0][0], res[0][0] / m * 1e3, res[1][0], bool(torch.equal(res[0][1], res[1][1]))))
