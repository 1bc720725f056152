"""Run the weight gradient of 3 x 3 / stride-1 dense-conv layers N times in one arithmetic mode (for rocprofv3):
    python tools/run_wgrad_layer.py [substring] [reps] [mode]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detmatch_amd import dense_conv
from bench_dense_conv import LAYERS
want = sys.argv[1] if len(sys.argv) > 1 else ''
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else 'fp32_split'
dev = torch.device('cuda:0')
dense_conv.set_math(mode)
for name, xs, cout, k, s, p in LAYERS:
    if want not in name or xs[1] % 4 or k != 3 or s != 1:
        continue
    n, cin, h, w = xs
    x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((n, cout, h, w), device=dev).contiguous(memory_format=torch.channels_last)
    dw = torch.empty(cout, cin, 3, 3, device=dev)
    taps = [(a - 1, b - 1) for a in range(3) for b in range(3)]
    for _ in range(reps):
        dense_conv._wgrad(dy, x, dw, None, [n, h, w, cout, cin, h, w, 1, 1, 9], taps, cin, cin * 9, 9, 1)
    torch.cuda.synchronize()
