"""Two layers (fwd + bwd) for rocprofv3 (kernel trace / PMC): python3 tools/prof_dense_conv.py [zeros]
`zeros`: all-zero operands — the chip holds a higher clock on them (MI355X_MICROARCH.md, DVFS
give-back), so the time difference to random data tells how much of a kernel's time is the clock."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
from detmatch_amd import dense_conv

dev = torch.device('cuda:0')
zeros = 'zeros' in sys.argv
for xs, cout in (((2, 128, 200, 176), 128), ((2, 256, 96, 320), 256)):
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, xs[1], 3, 3, device=dev) * 0.05)
    if zeros:
        x.zero_()
        w.data.zero_()
    xg = x.clone().requires_grad_(True)
    for _ in range(30):
        y = dense_conv.conv2d(xg, w, None, 1, 1)
        y.backward(torch.zeros_like(y) if zeros else torch.ones_like(y))
torch.cuda.synchronize()
