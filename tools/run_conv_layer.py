"""Run dense-conv layers' forward N times each (for rocprofv3 --kernel-trace / --pmc): python tools/run_conv_layer.py [substring] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detmatch_amd import dense_conv
from bench_dense_conv import LAYERS
want = sys.argv[1] if len(sys.argv) > 1 else ''
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
for name, xs, cout, k, s, p in LAYERS:
    if want not in name or xs[1] % 32:
        continue
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, xs[1], k, k, device=dev) * 0.05)
    with torch.no_grad():
        for _ in range(reps):
            dense_conv.conv2d(x, w, None, s, p)
    torch.cuda.synchronize()
