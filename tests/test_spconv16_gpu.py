"""Sparse gather-GEMM on the 16-bit matrix instructions (csrc/spconv16.hip) against the fp32 oracle at the
tolerance the storage type gives:

  * half-precision STORAGE (torch.float16 — the reference's indice_conv_half / indice_conv_backward_half,
    mmdet3d/ops/spconv/src/all.cc:35-36 — and torch.bfloat16): inputs are exactly representable, products exact,
    fp32 accumulation, ONE rounding of the result -> within 1 ulp of the storage type of the oracle's fp32 result;
  * the mixed-precision mode (precision.set_mixed: fp32 rows, bf16 multiplicands): equal to the oracle run on the
    bf16-ROUNDED operands to fp32 accuracy (2e-5), and within bf16 accuracy (2e-2 of the row scale) of exact fp32.
"""
import numpy as np
import pytest
import torch

from test_spconv_gpu import _rand_indices

pytestmark = pytest.mark.gpu
SHAPE = [21, 200, 176]
ULP = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def _scene(dev, rng, n, subm):
    from detmatch_amd.spconv import ops
    idx = _rand_indices(rng, n, 2, [9, 24, 24])            # dense enough for ~5 pairs per output
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev), 2, [9, 24, 24], ks, st, pd, 1, subm)
    rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
    return idx, rb


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('cin,cout,subm', [(16, 16, True), (16, 32, False), (32, 32, True), (32, 64, False),
                                           (64, 64, True), (64, 128, False)])
def test_half_storage_forward_and_backward(orc, dev, dtype, cin, cout, subm):
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(cin * 7 + cout + int(subm))
    idx, rb = _scene(dev, rng, 5000, subm)
    n_in, n_out = rb.n_in, rb.n_out
    x = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).to(dev).to(dtype)
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)).to(dev).to(dtype)
    dy = torch.from_numpy(rng.standard_normal((n_out, cout)).astype(np.float32)).to(dev).to(dtype)
    p, num = rb.indice_pairs.cpu().numpy(), rb.indice_num.cpu().numpy()
    xf, wf, dyf = x.float().cpu().numpy(), w.float().cpu().numpy(), dy.float().cpu().numpy()
    y = ops.indice_conv(x, w.view(3, 3, 3, cin, cout), rb.indice_pairs, rb.indice_num, n_out, False, subm)
    assert y.dtype == dtype and y.shape == (n_out, cout)
    want = orc.indice_conv(xf, wf, p, num, n_out, subm=subm)
    scale = np.abs(want).max(1, keepdims=True) + 1e-3
    assert np.abs(y.float().cpu().numpy() - want).max() <= 1.01 * ULP[dtype] * np.abs(want).max()
    assert np.median(np.abs(y.float().cpu().numpy() - want) / scale) < ULP[dtype]
    dx, dw = ops.indice_conv_backward(x, w.view(3, 3, 3, cin, cout), dy, rb.indice_pairs, rb.indice_num, False, subm)
    assert dx.dtype == dtype and dw.dtype == dtype and dw.shape == (3, 3, 3, cin, cout)
    dx_w, dw_w = orc.indice_conv_backward(xf, wf, dyf, p, num, subm=subm)
    assert np.abs(dx.float().cpu().numpy() - dx_w).max() <= 1.01 * ULP[dtype] * np.abs(dx_w).max()
    assert np.abs(dw.float().cpu().numpy().reshape(27, cin, cout) - dw_w).max() <= 1.01 * ULP[dtype] * np.abs(dw_w).max() + 1e-4


@pytest.mark.parametrize('cin,cout,subm', [(16, 16, True), (32, 64, False), (64, 64, True), (64, 128, False)])
def test_mixed_mode_is_bf16_multiplicands_fp32_accumulate(orc, dev, cin, cout, subm):
    from detmatch_amd import precision
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(cin + 3 * cout)
    idx, rb = _scene(dev, rng, 6000, subm)
    x = torch.from_numpy(rng.standard_normal((rb.n_in, cin)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)).to(dev)
    dy = torch.from_numpy(rng.standard_normal((rb.n_out, cout)).astype(np.float32)).to(dev)
    p, num = rb.indice_pairs.cpu().numpy(), rb.indice_num.cpu().numpy()
    exact = orc.indice_conv(x.cpu().numpy(), w.cpu().numpy(), p, num, rb.n_out, subm=subm)
    rounded = orc.indice_conv(x.bfloat16().float().cpu().numpy(), w.bfloat16().float().cpu().numpy(), p, num,
                              rb.n_out, subm=subm)
    y32 = ops.indice_conv(x, w.view(3, 3, 3, cin, cout), rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    with precision.mixed_precision():
        assert precision.mixed() and precision.sparse_bf16()
        y = ops.indice_conv(x, w.view(3, 3, 3, cin, cout), rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        dx, dw = ops.indice_conv_backward(x, w.view(3, 3, 3, cin, cout), dy, rb.indice_pairs, rb.indice_num, False, subm)
    assert not precision.mixed()
    assert y.dtype == torch.float32
    top = np.abs(exact).max()
    assert np.abs(y.cpu().numpy() - rounded).max() <= 2e-5 * top            # fp32 accumulation
    assert np.abs(y.cpu().numpy() - exact).max() <= 2e-2 * top              # bf16 multiplicands
    assert np.abs(y32.cpu().numpy() - exact).max() <= 1e-5 * top            # the default stays exact fp32
    assert float((y - y32).abs().max()) > 0                                 # ... and the switch does switch
    dx_w, dw_w = orc.indice_conv_backward(x.cpu().numpy(), w.cpu().numpy(), dy.cpu().numpy(), p, num, subm=subm)
    assert np.abs(dx.cpu().numpy() - dx_w).max() <= 2e-2 * np.abs(dx_w).max()
    assert np.abs(dw.cpu().numpy().reshape(27, cin, cout) - dw_w).max() <= 1e-4 * np.abs(dw_w).max()   # fp32 kernel


def test_half_precision_module_path(dev):
    """A SubMConv3d + SparseConv3d stack fed with half features / half weights (what the reference reaches with
    `.half()` modules) runs forward and backward in the storage type."""
    from detmatch_amd import spconv
    rng = np.random.default_rng(1)
    idx = _rand_indices(rng, 4000, 1, [9, 24, 24])
    net = spconv.SparseSequential(
        spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key='a'),
        spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False, indice_key='b')).to(dev).half()
    x = torch.randn(4000, 16, device=dev).half().requires_grad_(True)
    y = net(spconv.SparseConvTensor(x, torch.from_numpy(idx).to(dev), [9, 24, 24], 1))
    assert y.features.dtype == torch.float16 and torch.isfinite(y.features).all()
    y.features.float().square().mean().backward()
    assert x.grad.dtype == torch.float16 and float(x.grad.float().abs().sum()) > 0
    assert net[0].weight.grad.dtype == torch.float16 and torch.isfinite(net[1].weight.grad).all()


@pytest.mark.parametrize('cin,cout,subm', [(16, 16, True), (16, 32, False), (32, 32, True), (32, 64, False),
                                           (64, 64, True), (64, 128, False)])
def test_fp32_split_mode_matches_float64_better_than_the_fp32_instruction(dev, monkeypatch, cin, cout, subm):
    """DM_SP16_F32SPLIT: three-way bf16 split of both multiplicands, six products, fp32 accumulate.  Against a
    float64 evaluation of the same gather-GEMM its error must not exceed that of v_mfma_f32_16x16x4_f32
    (spconv_gr), forward and input gradient, and must sit at fp32 rounding level."""
    from detmatch_amd import precision
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(cin + 5 * cout)
    idx, rb = _scene(dev, rng, 6000, subm)
    x = torch.from_numpy(rng.standard_normal((rb.n_in, cin)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)).to(dev)
    dy = torch.from_numpy(rng.standard_normal((rb.n_out, cout)).astype(np.float32)).to(dev)
    # float64 reference from the gather table (pairs (in, out) per offset)
    p, num = rb.indice_pairs.cpu().numpy(), rb.indice_num.cpu().numpy()
    x64, w64, dy64 = x.cpu().double().numpy(), w.cpu().double().numpy(), dy.cpu().double().numpy()
    y_ref = np.zeros((rb.n_out, cout))
    dx_ref = np.zeros((rb.n_in, cin))
    for k in range(27):
        i, o = p[k, 0, :num[k]], p[k, 1, :num[k]]
        np.add.at(y_ref, o, x64[i] @ w64[k])
        np.add.at(dx_ref, i, dy64[o] @ w64[k].T)
    errs = {}
    for mode in ('fp32_mfma', 'fp32_split'):
        monkeypatch.setattr(precision, 'SPARSE_FP32', mode)
        y = ops.indice_conv(x, w.view(3, 3, 3, cin, cout), rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        dx, _ = ops.indice_conv_backward(x, w.view(3, 3, 3, cin, cout), dy, rb.indice_pairs, rb.indice_num, False, subm)
        errs[mode] = [float(np.sqrt(((got.cpu().double().numpy() - ref) ** 2).mean() / (ref ** 2).mean()))
                      for got, ref in ((y, y_ref), (dx, dx_ref))]
    for i in range(2):
        assert errs['fp32_split'][i] <= 4e-7, errs
        assert errs['fp32_split'][i] <= 1.05 * errs['fp32_mfma'][i] + 1e-8, errs


def test_pruned_spconv_variant_is_refused(dev):
    """The 32-row-tile fp32 variant (dm_spconv_set_variant(2)) lost its A/B and was removed in round 5: the switch
    refuses it instead of silently selecting something else."""
    from detmatch_amd import _lib
    L = _lib.lib()
    try:
        assert L.dm_spconv_set_variant(2) != 0
        assert L.dm_spconv_set_variant(1) == 0
    finally:
        L.dm_spconv_set_variant(-1)
