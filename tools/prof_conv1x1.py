"""One K-split 1 x 1 layer (ResNet stage 3, 1024 -> 256 at 4 x 24 x 78) forward, fp32-class split arithmetic, 40 launches — for
rocprofv3 --pmc / --kernel-trace:   python3 tools/prof_conv1x1.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
from detmatch_amd import dense_conv

dev = torch.device('cuda:0')
dense_conv.set_math('fp32_split')
for xs, cout in (((4, 1024, 24, 78), 256), ((4, 256, 24, 78), 1024)):
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, xs[1], 1, 1, device=dev) * 0.05)
    with torch.no_grad():
        for _ in range(40):
            y = dense_conv.conv2d(x, w, None, 1, 0)
torch.cuda.synchronize()
