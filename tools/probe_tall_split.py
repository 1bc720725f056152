"""Split choice of the batched BLAS weight gradient of the tall-skinny linears: time per split count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import detmatch_amd  # noqa
dev = torch.device('cuda', 0)


def t(fn):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best


for rows, cin, cout in ((884736, 132, 64), (884736, 64, 64), (131072, 68, 64), (131072, 64, 64), (65536, 68, 64), (131072, 36, 32), (65536, 16, 16)):
    x = torch.randn(rows, cin, device=dev); gy = torch.randn(rows, cout, device=dev)
    out = ['%7d %3d %3d |' % (rows, cin, cout)]
    for split in (1, 8, 16, 32, 64, 128, 256, 512, 1024):
        if rows % split or rows // split < 256:
            continue
        if split == 1:
            us = t(lambda: gy.t() @ x)
        else:
            us = t(lambda: torch.bmm(gy.view(split, rows // split, -1).transpose(1, 2), x.view(split, rows // split, -1)).sum(dim=0))
        out.append('%d: %.0f' % (split, us))
    print('  '.join(out))
