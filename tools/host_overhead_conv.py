"""Host time per dense_conv call (tiny problem: the device is never the bottleneck)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
import torch.nn.functional as F
from detmatch_amd import dense_conv
dev = torch.device('cuda:0')
x = torch.randn(1, 64, 8, 8, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
xg = x.clone().requires_grad_(True)
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('dense_conv fwd (no grad)   %.1f us/call' % t(lambda: dense_conv.conv2d(x, w.detach(), None, 1, 1)))
print('torch conv fwd (no grad)   %.1f us/call' % t(lambda: F.conv2d(x, w.detach(), None, 1, 1)))
def fb():
    y = dense_conv.conv2d(xg, w, None, 1, 1); y.backward(y)
def fbt():
    y = F.conv2d(xg, w, None, 1, 1); y.backward(y)
print('dense_conv fwd+bwd         %.1f us/call' % t(fb, 500))
print('torch conv fwd+bwd         %.1f us/call' % t(fbt, 500))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fb()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
