"""csrc/fc_gemm.hip (dm_fc_gemm forms 0 / 1 / 2) against float64 matrix products: the FC shapes of the step, ragged
shapes (rows / columns / contraction not multiples of the tile), unaligned leading dimensions, split contraction;
_lib.fc_linear's gradients against torch's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

SHAPES = [(256, 27648, 256), (200, 27648, 256), (256, 256, 256), (256, 256, 1), (256, 256, 7), (4096, 640, 128),
          (4096, 128, 256), (1024, 1024, 16), (1, 4, 1), (65, 33, 67), (130, 1000, 3), (1000, 36, 260), (3, 700, 5)]


def _err(got, want):
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('m,k,n', SHAPES)
def test_fc_gemm_three_forms_match_float64(dev, m, k, n):
    from detmatch_amd import _lib
    g = torch.Generator().manual_seed(m * 7 + k * 3 + n)
    x = torch.randn(m, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    gy = torch.randn(m, n, generator=g).to(dev)
    xd, wd, bd, gd = x.double(), w.double(), b.double(), gy.double()
    y = torch.empty(m, n, device=dev)
    _lib._fc_gemm(0, x, w, b, y, m, n, k, k, k, relu=False)
    assert _err(y, xd @ wd.t() + bd) < 3e-6
    _lib._fc_gemm(0, x, w, None, y, m, n, k, k, k, relu=True)
    assert _err(y, torch.relu(xd @ wd.t())) < 3e-6
    gx = torch.empty(m, k, device=dev)
    _lib._fc_gemm(1, gy, w, None, gx, m, k, n, n, k)
    assert _err(gx, gd @ wd) < 3e-6
    gw = torch.empty(n, k, device=dev)
    _lib._fc_gemm(2, gy, x, None, gw, n, k, m, n, k)
    assert _err(gw, gd.t() @ xd) < 3e-6
    # same inputs, same result: the split contraction is summed in a fixed order
    gw2 = torch.empty(n, k, device=dev)
    _lib._fc_gemm(2, gy, x, None, gw2, n, k, m, n, k)
    assert torch.equal(gw, gw2)


def test_fc_linear_gradients_equal_torch(dev):
    from detmatch_amd import _lib
    torch.manual_seed(3)
    for m, k, n, bias, relu in ((256, 27648, 256, False, False), (256, 256, 7, True, False), (512, 640, 128, True, True),
                                (2, 5, 8, 12, False) if False else (77, 100, 12, True, True)):
        x = torch.randn(m, k, device=dev, requires_grad=True)
        w = torch.nn.Parameter(torch.randn(n, k, device=dev) / k ** 0.5)
        b = torch.nn.Parameter(torch.randn(n, device=dev)) if bias else None
        g = torch.randn(m, n, device=dev)
        want = F.linear(x, w, b)
        want = torch.relu(want) if relu else want
        gw = torch.autograd.grad(want, [x, w] + ([b] if bias else []), g)
        calls = _lib.FC_GEMM_CALLS[0]
        got = _lib.blas_linear(x, w, b, relu=relu)
        gg = torch.autograd.grad(got, [x, w] + ([b] if bias else []), g)
        assert _lib.FC_GEMM_CALLS[0] == calls + 3
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
        for a, c in zip(gg, gw):
            assert float((a - c).abs().max()) <= 5e-5 * float(c.abs().max())
    # 3-D input (the shared MLP's (1, rows, C) view)
    x = torch.randn(2, 50, 64, device=dev)
    w = torch.randn(32, 64, device=dev)
    assert float((_lib.blas_linear(x, w) - F.linear(x, w)).abs().max()) < 1e-4
