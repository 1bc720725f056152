"""Weight gradient of the tall-skinny linears (dW = dy^T x over 10^5..10^6 rows) of one DetMatch iteration:
the shapes the step really issues, timed as (a) batched split-K BLAS + sum and (b) csrc/conv2d.hip's streaming
kernel (dm_tall_wgrad).
    python tools/bench_tall_wgrad.py
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import detmatch_amd  # noqa: F401
from detmatch_amd import dense_conv, pointnet2_stack as p2


def main():
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    dev = torch.device('cuda', 0)
    shapes = collections.Counter()
    orig = p2.TallSkinnyLinear.backward

    def spy(ctx, gy):
        x, w = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            shapes[(x.shape[0], x.shape[1], w.shape[0])] += 1
        return orig(ctx, gy)
    p2.TallSkinnyLinear.backward = staticmethod(spy)
    wl = DetMatchTrainWorkload(2, dev)
    wl.step()
    shapes.clear()
    wl.step()
    torch.cuda.synchronize()
    p2.TallSkinnyLinear.backward = staticmethod(orig)
    del wl

    def t(fn):
        for _ in range(3):
            fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        return best
    tot = [0.0, 0.0, 0.0]
    print('%9s %5s %5s %5s | %9s %9s | %8s  %s' % ('rows', 'cin', 'cout', 'calls', 'blas us', 'own us', 'HBM us', 'rel diff'))
    for (rows, cin, cout), calls in sorted(shapes.items(), key=lambda kv: -kv[0][0] * (kv[0][1] + kv[0][2]) * kv[1]):
        x = torch.randn(rows, cin, device=dev)
        gy = torch.randn(rows, cout, device=dev)
        split = next((s for s in (256, 128, 64, 32, 16, 8) if rows % s == 0 and rows // s >= 2048), 1)

        def blas():
            if split > 1:
                return torch.bmm(gy.view(split, rows // split, -1).transpose(1, 2), x.view(split, rows // split, -1)).sum(dim=0)
            return gy.t() @ x
        dw = torch.empty(cout, cin, device=dev)

        def own():
            return p2.TallSkinnyLinear._wgrad(gy, x)
        ok = cin % 4 == 0 and cout % 4 == 0
        a = t(blas)
        b = t(own) if ok else float('nan')
        ref = gy.double().t() @ x.double()
        d = float((own().double() - ref).abs().max() / ref.abs().max()) if ok else float('nan')
        d0 = float((blas().double() - ref).abs().max() / ref.abs().max())
        floor = rows * (cin + cout) * 4 / 8e12 * 1e6
        tot[0] += a * calls
        tot[1] += (b if ok else a) * calls
        tot[2] += floor * calls
        print('%9d %5d %5d %5d | %9.1f %9.1f | %8.1f  own %.1e blas %.1e' % (rows, cin, cout, calls, a, b, floor, d, d0))
    print('per step: blas %.0f us, own %.0f us, bytes at 8 TB/s %.0f us' % tuple(tot))


if __name__ == '__main__':
    main()
