"""Hand-written fp32-MFMA dense convolutions (csrc/conv2d.hip, detmatch_amd/dense_conv.py) vs
torch's convolution evaluated in float64 on the CPU: forward, input gradient, weight gradient, bias
gradient, for every layer TYPE of the DetMatch path (BEV backbone base_bev_backbone.py:38-69,
anchor-head 1x1, ResNet-50 caffe / FPN / RPN at split_0.py:39-99) at reduced spatial sizes.
Tolerance: 1e-4 relative to the largest reference magnitude (fp32 products, fp32 accumulation over
K = 27 ... 4608 terms; VERDICT r1 item 5 asks <= 1e-4 rel)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(got, want, what, tol=1e-4):
    want = want.to(torch.float64)
    scale = float(want.abs().max()) + 1e-12
    err = float((got.detach().cpu().to(torch.float64) - want).abs().max())
    assert err <= tol * scale, '%s: max err %.3e vs scale %.3e' % (what, err, scale)


CASES = [
    # name, (N, Cin, H, W), Cout, k, stride, pad, bias
    ('bev 3x3 128->128', (2, 128, 40, 36), 128, 3, 1, 1, False),
    ('bev first 3x3 256->128', (2, 256, 24, 20), 128, 3, 1, 1, False),
    ('bev 3x3 s2 128->256 odd', (2, 128, 41, 37), 256, 3, 2, 1, False),
    ('bev 3x3 256->256', (1, 256, 25, 22), 256, 3, 1, 1, False),
    ('resnet 1x1 s2 256->512', (2, 256, 24, 40), 512, 1, 2, 0, False),
    ('resnet 1x1 64->256', (2, 64, 24, 40), 256, 1, 1, 0, False),
    ('resnet 3x3 64->64', (2, 64, 24, 40), 64, 3, 1, 1, False),
    ('resnet 3x3 512->512 tiny', (2, 512, 6, 10), 512, 3, 1, 1, False),
    ('resnet 1x1 2048->512', (1, 2048, 6, 10), 512, 1, 1, 0, False),
    ('stem 7x7 s2 3->64', (2, 3, 64, 96), 64, 7, 2, 3, False),
    ('fpn 3x3 256->256 bias', (2, 256, 12, 20), 256, 3, 1, 1, True),
    ('fpn lateral 1x1 1024->256 bias', (2, 1024, 6, 10), 256, 1, 1, 0, True),
    ('heads 1x1 512->72 bias', (2, 512, 20, 22), 72, 1, 1, 0, True),
    ('rpn 1x1 256->16 bias', (2, 256, 12, 20), 16, 1, 1, 0, True),
    ('ragged 3x3 s2 p0 32->36', (3, 32, 17, 13), 36, 3, 2, 0, True),
]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv2d_forward_backward(dev, case):
    from detmatch_amd import dense_conv
    name, xs, cout, k, s, p, has_bias = case
    g = torch.Generator().manual_seed(abs(hash(name)) % 1000)
    x = torch.randn(xs, generator=g)
    w = torch.randn((cout, xs[1], k, k), generator=g) / np.sqrt(xs[1] * k * k)
    b = torch.randn(cout, generator=g) if has_bias else None
    xd = x.to(dev).requires_grad_(xs[1] >= 4)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if has_bias else None
    y = dense_conv.conv2d(xd, wd, bd, s, p)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if has_bias else None
    y64 = F.conv2d(x64, w64, b64, s, p)
    assert tuple(y.shape) == tuple(y64.shape)
    _close(y, y64, 'forward')
    dy = torch.randn(y64.shape, generator=g)
    y.backward(dy.to(dev))
    y64.backward(dy.double())
    _close(wd.grad, w64.grad, 'weight grad')
    if xs[1] >= 4:
        _close(xd.grad, x64.grad, 'input grad')
    if has_bias:
        _close(bd.grad, b64.grad, 'bias grad')


def test_frozen_bn_fold_and_fused_relu(dev):
    """conv(x, w * s) + b with ReLU in the epilogue (the 2D backbone's frozen BatchNorm fold,
    mm2d/backbone.py): gradients w.r.t. the UNSCALED weight."""
    from detmatch_amd import dense_conv
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 64, 20, 28), generator=g)
    w = torch.randn((128, 64, 3, 3), generator=g) / 24
    s = torch.rand(128, generator=g) + 0.5
    b = torch.randn(128, generator=g) * 0.1
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = dense_conv.conv2d(xd, wd, b.to(dev), 1, 1, relu=True, w_scale=s.to(dev))
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.relu(F.conv2d(x64, w64 * s.double().view(-1, 1, 1, 1), b.double(), 1, 1))
    _close(y, y64, 'forward')
    dy = torch.randn(y64.shape, generator=g)
    y.backward(dy.to(dev))
    y64.backward(dy.double())
    _close(wd.grad, w64.grad, 'weight grad')
    _close(xd.grad, x64.grad, 'input grad')


@pytest.mark.parametrize('k,cin,cout', [(1, 128, 256), (2, 256, 256)])
def test_conv_transpose2d(dev, k, cin, cout):
    """The BEV backbone's deblocks (base_bev_backbone.py:52-58): ConvTranspose2d k == stride."""
    from detmatch_amd import dense_conv
    g = torch.Generator().manual_seed(k)
    x = torch.randn((2, cin, 13, 11), generator=g)
    w = torch.randn((cin, cout, k, k), generator=g) / np.sqrt(cin)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = dense_conv.conv_transpose2d(xd, wd, k)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv_transpose2d(x64, w64, None, stride=k)
    assert tuple(y.shape) == tuple(y64.shape)
    _close(y, y64, 'forward')
    dy = torch.randn(y64.shape, generator=g)
    y.backward(dy.to(dev))
    y64.backward(dy.double())
    _close(wd.grad, w64.grad, 'weight grad')
    _close(xd.grad, x64.grad, 'input grad')


def test_weight_gradient_is_reproducible_and_cache_invalidates(dev):
    from detmatch_amd import dense_conv
    torch.manual_seed(0)
    x = torch.randn(2, 128, 30, 30, device=dev)
    conv = dense_conv.Conv2d(128, 128, 3, padding=1, bias=False).to(dev)
    grads = []
    for _ in range(2):
        conv.weight.grad = None
        conv(x).square().sum().backward()
        grads.append(conv.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])           # split-K with a fixed-order reduce
    y0 = conv(x)
    with torch.no_grad():
        conv.weight.mul_(2.0)                          # _version bump -> packed copy refreshed
    assert torch.allclose(conv(x), 2 * y0, rtol=1e-5, atol=1e-5)
    conv.weight.data.view(-1)[:] *= 0.5                # also bumps; raw-pointer writers call:
    dense_conv.weights_changed()
    assert torch.allclose(conv(x), y0, rtol=1e-5, atol=1e-5)


def test_all_packed_weights_refresh_in_one_launch(dev):
    """After `weights_changed()` (fused optimizer / EMA wrote through raw pointers) the first
    convolution re-packs EVERY cached weight of the stream with one dm_dconv_pack_batch launch:
    forward and input-gradient packings, scaled (frozen-BN fold) and plain, of several layers."""
    import ctypes
    from detmatch_amd import _lib, dense_conv
    torch.manual_seed(1)
    convs = [dense_conv.Conv2d(c_in, c_out, k, padding=k // 2, bias=False).to(dev)
             for c_in, c_out, k in ((16, 32, 3), (32, 64, 1), (64, 24, 3))]
    scale = torch.rand(24, device=dev) + 0.5
    scale.dm_constant = True

    def run(x):
        x = x.clone().requires_grad_(True)
        h = convs[1](convs[0](x))
        y = dense_conv.conv2d(h, convs[2].weight, None, 1, 1, relu=True, w_scale=scale)
        y.square().sum().backward()
        return y.detach(), x.grad.clone()

    def ref(x):
        x = x.double().clone().requires_grad_(True)
        h = torch.nn.functional.conv2d(x, convs[0].weight.double(), padding=1)
        h = torch.nn.functional.conv2d(h, convs[1].weight.double())
        y = torch.relu(torch.nn.functional.conv2d(h, convs[2].weight.double() * scale.double()[:, None, None, None],
                                                  padding=1))
        y.square().sum().backward()
        return y.detach(), x.grad.clone()

    x = torch.randn(2, 16, 20, 24, device=dev)
    run(x)                                              # fills the cache (one pack launch per weight)
    calls = {'single': 0, 'batch': 0}
    L = _lib.lib()
    real_single, real_batch = L.dm_dconv_pack, L.dm_dconv_pack_batch

    class Counting(object):
        def __getattr__(self, name):
            fn = getattr(L, name)
            if name in ('dm_dconv_pack', 'dm_dconv_pack_batch'):
                def counted(*a):
                    calls['single' if name == 'dm_dconv_pack' else 'batch'] += 1
                    return fn(*a)
                return counted
            return fn
    orig = _lib.lib
    _lib.lib = lambda: Counting()
    try:
        for step in range(3):
            with torch.no_grad():
                for c in convs:
                    c.weight.data.view(-1)[:] *= 1.1       # raw write: no Tensor._version bump on the Parameter
            dense_conv.weights_changed()
            y, gx = run(x)
            yr, gxr = ref(x)
            assert torch.allclose(y.double(), yr, rtol=1e-4, atol=1e-4)
            assert torch.allclose(gx.double(), gxr, rtol=1e-4, atol=1e-3)
    finally:
        _lib.lib = orig
    assert calls == {'single': 0, 'batch': 3}, calls


def test_cpu_tensors_are_refused(monkeypatch):
    from detmatch_amd import _lib, dense_conv
    monkeypatch.setattr(dense_conv, 'HOST_TENSOR_HOOK', None)   # the product's setting
    with pytest.raises(_lib.DetMatchHipError):
        dense_conv.conv2d(torch.zeros(1, 4, 8, 8), torch.zeros(4, 4, 3, 3), None, 1, 1)


@pytest.mark.parametrize('shape', [
    ((2, 512, 12, 40), 512, 3, 1, 1),      # ResNet layer4 3x3: 120 tiles of 64x64 -> split-K
    ((2, 1024, 24, 80), 256, 1, 1, 0),     # layer3 1x1
    ((2, 256, 6, 20), 256, 3, 1, 1),       # FPN P5 3x3: 8 tiles
    ((2, 256, 100, 88), 256, 3, 1, 1),     # BEV block2: 1100 tiles of 64x64 = one round + tail
    ((1, 64, 200, 176), 128, 3, 1, 1),     # 550 x 2 tiles
], ids=['splitk l4', 'splitk l3 1x1', 'splitk P5', 'round+tail', 'round+tail 2'])
def test_scheduling_paths_agree_with_reference(dev, shape):
    """Every launch plan of dm_dconv_gemm (split-K with reduce, full rounds + limited tail) against
    the float64 convolution, plus bitwise run-to-run reproducibility."""
    from detmatch_amd import dense_conv
    xs, cout, k, s, p = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(xs, generator=g)
    w = torch.randn((cout, xs[1], k, k), generator=g) / np.sqrt(xs[1] * k * k)
    b = torch.randn(cout, generator=g)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = dense_conv.conv2d(xd, wd, bd, s, p, relu=True)
    y64 = F.relu(F.conv2d(x.double(), w.double(), b.double(), s, p))
    _close(y, y64, 'forward')
    assert torch.equal(y, dense_conv.conv2d(xd, wd, bd, s, p, relu=True))


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize('shape', [
    ((2, 128, 100, 88), 128, 3, 1, 1),      # 128x128 rounds + 64x64 tail
    ((2, 128, 60, 44), 256, 3, 2, 1),       # stride 2: input gradient per residue class
    ((2, 512, 12, 40), 512, 3, 1, 1),       # split-K
    ((2, 1024, 24, 80), 256, 1, 1, 0),      # 1x1
    ((1, 64, 96, 160), 64, 3, 1, 1),        # Cin = 64: one K-tile per tap
    ((2, 256, 30, 30), 72, 1, 1, 0),        # Cout not a multiple of the tile
], ids=['3x3 128', '3x3 s2', 'splitk', '1x1', 'cin64', 'cout72'])
def test_mixed_precision_math_mode(dev, shape):
    """dm_dconv_set_math(1): bf16 multiplicands, fp32 accumulation.  Forward and input gradient equal the
    float64 convolution of the bf16-ROUNDED operands to fp32-accumulation accuracy, and the exact
    convolution to bf16 accuracy (what the reference's fp16 autocast configs accept); the weight
    gradient likewise for layers with more than 64 channels on both sides (the others stay fp32)."""
    from detmatch_amd import dense_conv
    xs, cout, k, s, p = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(xs, generator=g)
    w = torch.randn(cout, xs[1], k, k, generator=g) / (xs[1] * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    gy_seed = 22
    assert dense_conv.get_math() in ('fp32_mfma', 'fp32_split')
    dense_conv.set_math('bf16')
    try:
        xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = torch.nn.Parameter(w.to(dev))
        y = dense_conv.conv2d(xd, wd, b.to(dev), s, p)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(gy_seed))
        y.backward(gy.to(dev))
        assert dense_conv.get_math() == 'bf16'
    finally:
        dense_conv.set_math('fp32')
    F = torch.nn.functional
    # (a) rounded operands, exact arithmetic
    x64, w64 = _bf16_round(x).requires_grad_(True), _bf16_round(w)
    yr = F.conv2d(x64, w64, b.double(), s, p)
    scale = float(yr.abs().max())
    assert float((y.detach().cpu().double() - yr).abs().max()) <= 2e-5 * scale
    gxr = torch.autograd.grad(F.conv2d(torch.zeros_like(x64).requires_grad_(True), w64, None, s, p).sum() * 0 +
                              (F.conv2d(x64, w64, None, s, p) * _bf16_round(gy)).sum(), x64)[0]
    gscale = float(gxr.abs().max())
    # the input gradient reduces over Cout: it takes the bf16 kernel when Cout % 64 == 0, else stays fp32
    tol = 2e-5 if cout % 64 == 0 else 2e-2
    assert float((xd.grad.cpu().double() - gxr).abs().max()) <= tol * gscale
    # (b) the exact convolution, to half-precision accuracy
    ye = F.conv2d(x.double(), w.double(), b.double(), s, p)
    assert float((y.detach().cpu().double() - ye).abs().max()) <= 2e-2 * scale
    # (c) weight gradient: bf16 multiplicands when both channel counts exceed 64, fp32 arithmetic otherwise
    mixed_w = xs[1] > 64 and cout > 64
    xe = (_bf16_round(x) if mixed_w else x.double())
    ge = (_bf16_round(gy) if mixed_w else gy.double())
    we = w.double().requires_grad_(True)
    (F.conv2d(xe, we, None, s, p) * ge).sum().backward()
    assert float((wd.grad.cpu().double() - we.grad).abs().max()) <= 1e-4 * float(we.grad.abs().max())
    if mixed_w:      # and the exact gradient to half-precision accuracy
        we2 = w.double().requires_grad_(True)
        (F.conv2d(x.double(), we2, None, s, p) * gy.double()).sum().backward()
        assert float((wd.grad.cpu().double() - we2.grad).abs().max()) <= 2e-2 * float(we2.grad.abs().max())


@pytest.mark.parametrize('shape', [((2, 64, 24, 40), 256), ((1, 512, 6, 10), 2048), ((2, 128, 13, 21), 512),
                                   ((2, 32, 9, 11), 36)],
                         ids=['64->256', 'split-K 512->2048', 'odd 128->512', 'ragged 32->36'])
@pytest.mark.parametrize('math', ['fp32', 'bf16'])
def test_residual_epilogue_is_add_then_relu(dev, shape, math):
    """relu(conv(x) * s + b + identity) in the GEMM's epilogue (the last 1x1 of a ResNet bottleneck with
    its frozen BatchNorm) == the same from separate kernels: outputs bit for bit (same accumulator, same
    fp32 add), gradients w.r.t. input, weight and identity equal."""
    from detmatch_amd import dense_conv
    (n, cin, h, w), cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((n, cin, h, w), generator=g).to(dev)
    wt = (torch.randn((cout, cin, 1, 1), generator=g) / np.sqrt(cin)).to(dev)
    scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
    shift = torch.randn(cout, generator=g).to(dev)
    idt = torch.randn((n, cout, h, w), generator=g).to(dev)
    dy = torch.randn((n, cout, h, w), generator=g).to(dev)
    prev = dense_conv.get_math()
    dense_conv.set_math(math)
    try:
        res = []
        for fused in (True, False):
            xs, ws, ids = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), idt.clone().requires_grad_(True)
            if fused:
                y = dense_conv.conv2d(xs, ws, shift, 1, 0, relu=True, w_scale=scale, residual=ids)
            else:
                y = F.relu(dense_conv.conv2d(xs, ws, shift, 1, 0, w_scale=scale) + ids)
            y.backward(dy)
            res.append((y.detach(), xs.grad, ws.grad, ids.grad))
    finally:
        dense_conv.set_math(prev)
    a, b = res
    assert torch.equal(a[0], b[0])
    assert float((a[0] == 0).float().mean()) > 0.2          # the ReLU is active
    for u, v, what in zip(a[1:], b[1:], ('dx', 'dw', 'd identity')):
        assert torch.equal(u, v), what


@pytest.mark.parametrize('shape', [
    ((2, 128, 40, 44), 128, 3, 1, 1),       # 3x3, 128x128 tiles + 64x64 tail
    ((2, 128, 40, 44), 256, 3, 2, 1),       # strided (input gradient per residue class)
    ((2, 512, 12, 40), 512, 3, 1, 1),       # split-K
    ((2, 1024, 24, 80), 256, 1, 1, 0),      # 1x1
    ((2, 256, 30, 30), 72, 1, 1, 0),        # Cout not a multiple of the tile
    ((1, 32, 20, 24), 64, 3, 1, 1),         # Cin = 32
    ((2, 128, 104, 176), 128, 3, 1, 1),     # patch kernel (3x3 stride 1, >= 200 tiles): BN = 128
    ((2, 256, 100, 88), 256, 3, 1, 1),      # patch kernel: partial 8 x 16 tiles on both axes, two column tiles
    ((2, 64, 96, 160), 64, 3, 1, 1),        # patch kernel: BN = 64 (4 x 1 waves)
    ((1, 64, 200, 176), 72, 3, 1, 1),       # patch kernel: Cout not a multiple of the tile
], ids=['3x3 128', '3x3 s2', 'splitk', '1x1', 'cout72', 'cin32', 'patch128', 'patch partial', 'patch64', 'patch cout72'])
def test_fp32_split_mode_is_at_least_as_accurate_as_the_fp32_instruction(dev, shape):
    """dm_dconv_set_math(2): fp32-class arithmetic from six bf16 products of the three-way split operands.
    Against the float64 convolution its error must not exceed that of the matrix pipe's own fp32 instruction
    (measured: about a third of it on deep reductions, equal within the rounding noise on shallow ones — K = 576
    for the 64-channel layers), forward and both gradients — and must meet the fp32 bar of this file (1e-4 of the
    output scale) with two orders of magnitude to spare."""
    from detmatch_amd import dense_conv
    xs, cout, k, s, p = shape
    g = torch.Generator().manual_seed(31)
    x = torch.randn(xs, generator=g)
    w = torch.randn(cout, xs[1], k, k, generator=g) / (xs[1] * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    F = torch.nn.functional
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b.double(), s, p)
    gy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(32))
    y64.backward(gy.double())
    want = (y64.detach(), x64.grad, w64.grad)
    errs = {}
    prev = dense_conv.get_math()
    try:
        for mode in ('fp32_mfma', 'fp32_split'):
            dense_conv.set_math(mode)
            xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wd = torch.nn.Parameter(w.to(dev))
            y = dense_conv.conv2d(xd, wd, b.to(dev), s, p)
            y.backward(gy.to(dev))
            errs[mode] = [float((got.detach().cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
                          for got, ref in zip((y, xd.grad, wd.grad), want)]
    finally:
        dense_conv.set_math(prev)
    for i, what in enumerate(('forward', 'input gradient', 'weight gradient')):
        assert errs['fp32_split'][i] <= 1e-6, (what, errs)
        assert errs['fp32_split'][i] <= 1.25 * errs['fp32_mfma'][i] + 2e-8, (what, errs)


@pytest.mark.parametrize('shape', [(2, 128, 40, 48, 128), (1, 64, 33, 70, 64), (2, 256, 24, 80, 256), (2, 128, 200, 176, 128),
                                   (1, 96, 17, 19, 72)])
def test_planes_weights_are_bit_identical(dev, shape):
    """3 x 3 / stride-1 layers with the weights pre-split into bf16 planes and moved by LDS-DMA (dm_dconv_gemm_planes,
    dconv_patch_gl_kernel) against the in-kernel split of the same layers: same split, same product order, so the
    forward outputs and the input gradients are bit-identical (ragged image borders, column tiles hanging over
    Cout, the lattice tail behind the whole rounds included)."""
    from detmatch_amd import dense_conv
    b, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev))
    bias = torch.randn(cout, generator=g).to(dev)
    dy = torch.randn(b, cout, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    res = []
    prev = dense_conv.PLANES
    try:
        for planes in (True, False):
            dense_conv.PLANES = planes
            dense_conv._PACK_CACHE.clear()
            xg = x.clone().requires_grad_(True)
            y = dense_conv.conv2d(xg, wt, bias, 1, 1, relu=True)
            y.backward(dy)
            res.append((y.detach().clone(), xg.grad.clone(), wt.grad.clone()))
            wt.grad = None
    finally:
        dense_conv.PLANES = prev
        dense_conv._PACK_CACHE.clear()
    for a, b_ in zip(*res):
        assert torch.equal(a, b_)


def _both_modes(dev, x, w, gy, s, p):
    """(forward, input gradient, weight gradient) of the convolution in the native fp32 and in the split arithmetic."""
    from detmatch_amd import dense_conv
    out = {}
    prev = dense_conv.get_math()
    try:
        for mode in ('fp32_mfma', 'fp32_split'):
            dense_conv.set_math(mode)
            dense_conv._PACK_CACHE.clear()
            xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wd = torch.nn.Parameter(w.to(dev))
            y = dense_conv.conv2d(xd, wd, None, s, p)
            y.backward(gy.to(dev))
            out[mode] = [t.detach().cpu().double() for t in (y, xd.grad, wd.grad)]
    finally:
        dense_conv.set_math(prev)
        dense_conv._PACK_CACHE.clear()
    return out


_ADVERSARIAL = [((2, 128, 24, 32), 128, 3, 1, 1), ((2, 256, 20, 24), 64, 1, 1, 0), ((1, 64, 33, 40), 128, 3, 2, 1)]


@pytest.mark.parametrize('shape', _ADVERSARIAL, ids=['3x3 patch', '1x1', '3x3 s2'])
@pytest.mark.parametrize('kind', ['exponent spread 2^30', 'powers of two', 'tiny (FLT_MIN * 2^17 ...)'])
def test_fp32_split_max_error_on_adversarial_inputs(dev, shape, kind):
    """MAXIMUM error (not rms) of the split arithmetic against float64, element by element, in units of the natural
    scale of each output S = sum_k |x_k| |w_k| (what an fp32 accumulation of K products is measured against):
      * inputs whose magnitudes spread over 2^30 (per element random exponent in [-15, 15]);
      * exact powers of two (the split is exact: m = l = 0; only the fp32 accumulation rounds);
      * inputs just above the range where the LOWEST plane would underflow (|x| in FLT_MIN * [2^17, 2^20]): all three
        planes are normal numbers there, so full precision is kept (include/detmatch_hip.h states the range below it).
    The bound: the split's worst element is within 4 fp32 ulps (4 * 2^-24) of S per sqrt(K) accumulated products —
    and never worse than 1.5x the worst element of the matrix pipe's own fp32 instruction on the same inputs."""
    xs, cout, k, s, p = shape
    g = torch.Generator().manual_seed(77)
    K = xs[1] * k * k
    if kind.startswith('exponent'):
        x = torch.randn(xs, generator=g) * torch.exp2(torch.randint(-15, 16, xs, generator=g).float())
        w = torch.randn(cout, xs[1], k, k, generator=g) * torch.exp2(torch.randint(-15, 16, (cout, xs[1], k, k), generator=g).float())
    elif kind.startswith('powers'):
        x = torch.exp2(torch.randint(-8, 9, xs, generator=g).float()) * (torch.randint(0, 2, xs, generator=g).float() * 2 - 1)
        w = torch.exp2(torch.randint(-8, 9, (cout, xs[1], k, k), generator=g).float()) * \
            (torch.randint(0, 2, (cout, xs[1], k, k), generator=g).float() * 2 - 1)
    else:
        tiny = float(np.float32(1.17549435e-38)) * 2.0 ** 17
        x = (torch.rand(xs, generator=g) * 7 + 1) * tiny * (torch.randint(0, 2, xs, generator=g).float() * 2 - 1)
        w = torch.randn(cout, xs[1], k, k, generator=g)
    F = torch.nn.functional
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p)
    gy = torch.randn(y64.shape, generator=g)
    y64.backward(gy.double())
    # natural scales: the same three contractions of the absolute values
    xa, wa = x.double().abs().requires_grad_(True), w.double().abs().requires_grad_(True)
    ya = F.conv2d(xa, wa, None, s, p)
    ya.backward(gy.double().abs())
    scales = (ya.detach(), xa.grad, wa.grad)
    refs = (y64.detach(), x64.grad, w64.grad)
    depth = (K, cout * k * k, xs[0] * y64.shape[2] * y64.shape[3])      # products accumulated per output element
    res = _both_modes(dev, x, w, gy, s, p)
    ulp = 2.0 ** -24
    for i, what in enumerate(('forward', 'input gradient', 'weight gradient')):
        sc = scales[i].clamp_min(1e-300)
        e_split = float(((res['fp32_split'][i] - refs[i]).abs() / sc).max())
        e_mfma = float(((res['fp32_mfma'][i] - refs[i]).abs() / sc).max())
        bound = 4 * ulp * depth[i] ** 0.5
        assert e_split <= bound, (what, kind, e_split, bound, e_mfma)
        assert e_split <= 1.5 * e_mfma + ulp, (what, kind, e_split, e_mfma)


@pytest.mark.parametrize('shape', _ADVERSARIAL, ids=['3x3 patch', '1x1', '3x3 s2'])
def test_fp32_split_non_finite_inputs_stay_non_finite(dev, shape):
    """+-inf and NaN in the input: every output element the native fp32 instruction makes non-finite is non-finite
    in the split arithmetic too, and every other element is finite and equal to the clean result (an infinite x
    splits into h = inf, m = l = NaN, so the split answers NaN where the fp32 instruction answers +-inf: which kind
    of non-finite value is NOT preserved — include/detmatch_hip.h says so)."""
    xs, cout, k, s, p = shape
    g = torch.Generator().manual_seed(78)
    x = torch.randn(xs, generator=g)
    w = torch.randn(cout, xs[1], k, k, generator=g) / (xs[1] * k * k) ** 0.5
    gy = torch.randn((xs[0], cout, (xs[2] + 2 * p - k) // s + 1, (xs[3] + 2 * p - k) // s + 1), generator=g)
    clean = _both_modes(dev, x, w, gy, s, p)
    xb = x.clone()
    xb[0, 3, 5, 7] = float('inf')
    xb[0, 9, 11, 2] = float('-inf')
    xb[xs[0] - 1, 1, 2, 3] = float('nan')
    bad = _both_modes(dev, xb, w, gy, s, p)
    for i, what in enumerate(('forward', 'weight gradient')):
        j = 0 if i == 0 else 2                   # (the input gradient does not read x)
        nf_mfma = ~torch.isfinite(bad['fp32_mfma'][j])
        nf_split = ~torch.isfinite(bad['fp32_split'][j])
        assert nf_mfma.any()
        assert torch.equal(nf_mfma, nf_split), what
        ok = ~nf_split
        torch.testing.assert_close(bad['fp32_split'][j][ok], clean['fp32_split'][j][ok], rtol=0, atol=0)


@pytest.mark.parametrize('shape', [
    ((2, 128, 24, 32), 128),      # two full 16-pixel segments per row
    ((2, 64, 13, 25), 64),        # ragged last segment (25 = 16 + 9), odd height
    ((1, 256, 9, 8), 128),        # rows shorter than one segment
    ((3, 128, 7, 50), 72),        # dY channels not a multiple of 4 x 16; three images
    ((2, 72, 20, 44), 256),       # X channels not a multiple of the tile
], ids=['24x32', '13x25 ragged', '9x8 short', 'cout72', 'cin72'])
def test_tap_fused_weight_gradient_equals_per_tap_kernel(dev, shape):
    """3 x 3 / stride-1 weight gradients in the split arithmetic: dconv_wgrad9_kernel (all nine taps in one workgroup,
    the 3 x 18-pixel halo of X staged once) against (a) the per-tap kernel on the same inputs — same six products per
    term, another summation order over the pixels, so agreement to the fp32 accumulation error — and (b) float64.
    Also: bit-for-bit run-to-run reproducibility and accumulation into an existing gradient."""
    from detmatch_amd import dense_conv
    xs, cout = shape
    g = torch.Generator().manual_seed(91)
    x = torch.randn(xs, generator=g)
    w = torch.randn(cout, xs[1], 3, 3, generator=g) / (xs[1] * 9) ** 0.5
    gy = torch.randn((xs[0], cout, xs[2], xs[3]), generator=g)
    ref = F.conv2d(x.double(), w.double().requires_grad_(True), None, 1, 1)
    w64 = torch.nn.Parameter(w.double())
    F.conv2d(x.double(), w64, None, 1, 1).backward(gy.double())
    wa = torch.nn.Parameter(w.double().abs())
    F.conv2d(x.double().abs(), wa, None, 1, 1).backward(gy.double().abs())
    scale = wa.grad.clamp_min(1e-300)
    depth = xs[0] * xs[2] * xs[3]
    prev = dense_conv.get_math()
    got = {}
    try:
        for mode in ('fp32_split', 'fp32_split_tapwise_wgrad'):
            dense_conv.set_math(mode)
            dense_conv._PACK_CACHE.clear()
            runs = []
            for _ in range(2):
                xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                wd = torch.nn.Parameter(w.to(dev))
                dense_conv.conv2d(xd, wd, None, 1, 1).backward(gy.to(dev))
                runs.append(wd.grad.clone())
            assert torch.equal(runs[0], runs[1]), mode
            # a second backward accumulates into the existing gradient (the kernel's accumulate path)
            dense_conv.conv2d(xd, wd, None, 1, 1).backward(gy.to(dev))
            torch.testing.assert_close(wd.grad, 2 * runs[0], rtol=1e-6, atol=1e-6)
            got[mode] = runs[0].cpu().double()
    finally:
        dense_conv.set_math(prev)
        dense_conv._PACK_CACHE.clear()
    ulp = 2.0 ** -24
    for mode, v in got.items():
        err = float(((v - w64.grad).abs() / scale).max())
        assert err <= 4 * ulp * depth ** 0.5, (mode, err)
    d = float(((got['fp32_split'] - got['fp32_split_tapwise_wgrad']).abs() / scale).max())
    assert d <= 4 * ulp * depth ** 0.5, d
