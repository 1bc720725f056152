"""Host cost (us per call, device never the bottleneck) of the building blocks of a Python-issued step:
a ctypes launch through the C-ABI, torch.empty, a tiny aten op, a trivial autograd.Function."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
from detmatch_amd import _lib
dev = torch.device('cuda:0')
L = _lib.lib()
a = torch.zeros(64, device=dev); b = torch.zeros(64, device=dev)


def t(fn, n=5000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def show(name, fn, n=5000):
    h, d = t(fn, n)
    print('%-44s host %6.2f us/call   drained %6.2f us/call' % (name, h, d))


pa, pb, st = _lib.ptr(a), _lib.ptr(b), _lib.stream()
show('ctypes dm_ema_update_f32 (prebuilt args)', lambda: L.dm_ema_update_f32(pa, pb, 64, 0.5, st))
show('ctypes dm_ema_update_f32 (+ptr() +stream())', lambda: L.dm_ema_update_f32(_lib.ptr(a), _lib.ptr(b), 64, 0.5, _lib.stream()))
show('torch.empty(64)', lambda: torch.empty(64, device=dev))
show('a.add_(b)', lambda: a.add_(b))
show('a + b', lambda: a + b)
show('torch.zeros(64)', lambda: torch.zeros(64, device=dev))
show('a.view(8, 8)', lambda: a.view(8, 8))


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view(-1)

    @staticmethod
    def backward(ctx, g):
        return g


ag = a.clone().requires_grad_(True)
show('autograd.Function.apply (view only, grad on)', lambda: F.apply(ag))
with torch.no_grad():
    show('autograd.Function.apply (no_grad)', lambda: F.apply(ag))
def fb():
    y = F.apply(ag); y.backward(b)
show('Function fwd + backward()', fb, 2000)
