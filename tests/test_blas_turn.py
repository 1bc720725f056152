"""One vendor GEMM at a time (detmatch_amd/_lib.py:blas_turn) — host logic on the CPU: autograd of blas_linear equals
F.linear's, turns nest and are counted, threads exclude each other."""
import threading
import time

import torch
import torch.nn.functional as F

from detmatch_amd import _lib


def test_blas_linear_equals_linear_with_gradients():
    torch.manual_seed(0)
    for shape, bias in (((7, 12), True), ((3, 5, 12), False), ((1, 12), True)):
        x = torch.randn(*shape, requires_grad=True)
        w = torch.randn(9, 12, requires_grad=True)
        b = torch.randn(9, requires_grad=True) if bias else None
        y = _lib.blas_linear(x, w, b)
        g = torch.randn_like(y)
        got = torch.autograd.grad(y, [x, w] + ([b] if bias else []), g)
        x2, w2 = x.detach().requires_grad_(), w.detach().requires_grad_()
        b2 = b.detach().requires_grad_() if bias else None
        y2 = F.linear(x2, w2, b2)
        want = torch.autograd.grad(y2, [x2, w2] + ([b2] if bias else []), g)
        assert torch.equal(y, y2)
        for a, c in zip(got, want):
            assert torch.allclose(a, c, rtol=1e-5, atol=1e-6)
    # no grad needed: plain call, still inside a turn
    n0 = _lib.BLAS_TURNS[0]
    with torch.no_grad():
        _lib.blas_linear(torch.randn(2, 12), torch.randn(9, 12))
    assert _lib.BLAS_TURNS[0] == n0 + 1
    # a frozen weight gets no gradient, the input does
    x = torch.randn(4, 12, requires_grad=True)
    y = _lib.blas_linear(x, torch.randn(9, 12))
    y.sum().backward()
    assert x.grad is not None


def test_turns_exclude_other_threads_and_nest():
    order = []

    def other():
        with _lib.blas_turn():
            order.append('other')
    with _lib.blas_turn():
        with _lib.blas_turn():           # re-entrant (a GEMM helper called from inside a turn)
            pass
        t = threading.Thread(target=other)
        t.start()
        time.sleep(0.2)
        order.append('main')
    t.join()
    assert order == ['main', 'other']
