"""Chained sub-graphs == the op-by-op path, bit for bit (same kernels, same arguments, same order): outputs, input
gradients, parameter gradients, BatchNorm running statistics and call counters."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _bev_pair(dev, c_in=64, heads=True):
    from test_pcdet_torch_golden import ConfigDict, configs
    from detmatch_amd.pcdet.backbones_2d import BaseBEVBackbone
    from detmatch_amd.pcdet.dense_heads import AnchorHeadSingle
    torch.manual_seed(3)
    bb = BaseBEVBackbone(ConfigDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[64, 128],
                                    UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128]), input_channels=c_in)
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    grid = np.array([176 * 8 // 4, 200 * 8 // 4, 40])      # a quarter-size BEV grid: 50 x 44 feature map
    pcr = np.array([0, -40, -3, 70.4, 40, 1], dtype=np.float32)
    head = AnchorHeadSingle(cfg.DENSE_HEAD, input_channels=256, num_class=3, class_names=configs.CLASS_NAMES,
                            grid_size=grid, point_cloud_range=pcr)
    for m in bb.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.5, 2.0)
    bb, head = bb.to(dev), head.to(dev)
    bb.__dict__['fused_head'] = head
    return bb, head


def _run(bb, head, x, train, enabled):
    from detmatch_amd import chain
    old = chain.ENABLED
    chain.ENABLED = enabled
    try:
        bb.train(train), head.train(train)
        for p in list(bb.parameters()) + list(head.parameters()):
            p.grad = None
        x = x.clone().requires_grad_(train)
        with torch.set_grad_enabled(train):
            d = bb(dict(spatial_features=x))
            heads = d.get('dense_head_convs')
            if heads is None:
                heads = head.conv_heads(d['spatial_features_2d'])
        out = dict(heads=[h.detach().clone() for h in heads], f2d=d['spatial_features_2d'].detach().clone(),
                   levels={k: v.detach().clone() for k, v in d.items() if k.startswith('spatial_features_') and k[-1] == 'x'})
        if train:
            g = torch.Generator(device='cpu').manual_seed(1)
            loss = sum((h * torch.randn(h.shape, generator=g).to(h.device)).sum() for h in heads)
            loss.backward()
            out['gx'] = x.grad.clone()
            out['grads'] = {n: p.grad.clone() for n, p in list(bb.named_parameters()) + list(head.named_parameters())
                            if p.grad is not None}
        out['state'] = {k: v.clone() for k, v in bb.state_dict().items()}
        return out
    finally:
        chain.ENABLED = old


def _same(a, b, what):
    assert a.shape == b.shape, what
    assert torch.equal(a, b), '%s differs: max |d| = %g' % (what, float((a - b).abs().max()))


@pytest.mark.parametrize('train', [True, False])
def test_bev_backbone_chain_equals_op_by_op(dev, train):
    from detmatch_amd import bn_relu
    bb, head = _bev_pair(dev)
    state = copy.deepcopy(bb.state_dict())
    x = torch.randn(2, 64, 50, 44, device=dev).contiguous(memory_format=torch.channels_last)
    ref = _run(bb, head, x, train, enabled=False)
    bb.load_state_dict(state)
    got = _run(bb, head, x, train, enabled=True)
    assert bb.__dict__.get('_chains'), 'the chain did not run'
    ch = list(bb.__dict__['_chains'].values())[0][0]
    assert ch.launches()[0] >= 15
    for i, (a, b) in enumerate(zip(got['heads'], ref['heads'])):
        _same(a, b, 'head %d' % i)
    _same(got['f2d'], ref['f2d'], 'spatial_features_2d')
    assert set(got['levels']) == set(ref['levels'])
    for k in ref['levels']:
        _same(got['levels'][k], ref['levels'][k], k)
    for k in ref['state']:
        _same(got['state'][k], ref['state'][k], 'state ' + k)
    if train:
        _same(got['gx'], ref['gx'], 'input gradient')
        assert set(got['grads']) == set(ref['grads'])
        for k in ref['grads']:
            if k.endswith('.bias') and k.startswith('conv_'):
                # bias gradient = column sums of the output gradient: the chain's own two-stage fixed-order kernel
                # (dm_colsum_f32) instead of aten::sum — another summation order, same value to fp32 rounding
                a, b = got['grads'][k], ref['grads'][k]
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-6, 'grad ' + k
            else:
                _same(got['grads'][k], ref['grads'][k], 'grad ' + k)
    # a second call after the weights moved (in-place update: version counters) re-derives the packed copies
    with torch.no_grad():
        for p in bb.parameters():
            p.mul_(1.01)
    state2 = copy.deepcopy(bb.state_dict())
    got2 = _run(bb, head, x, train, enabled=True)
    bb.load_state_dict(state2)
    ref2 = _run(bb, head, x, train, enabled=False)
    _same(got2['f2d'], ref2['f2d'], 'spatial_features_2d after an update')
    assert not torch.equal(got2['f2d'], got['f2d'])


def test_chain_refuses_cpu_tensors():
    from detmatch_amd import _lib, chain
    p = chain.Program('x')
    with pytest.raises(_lib.DetMatchHipError):
        p.call('dm_relu_mask_f32', torch.zeros(4), torch.zeros(4), torch.zeros(4), 4, chain.Program.STREAM)


def _frcnn(dev):
    from detmatch_amd import configs
    from detmatch_amd.mm2d import FasterRCNN
    torch.manual_seed(0)
    cfg = configs.frcnn_kitti_model()
    cfg.pop('type')
    m = FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg).to(dev)
    with torch.no_grad():      # zero_init_residual zeroes bn3.weight: make every branch carry signal
        for mod in m.backbone.modules():
            if type(mod).__name__ == 'FrozenBN':
                mod.weight.uniform_(0.5, 1.0)
                mod.bias.uniform_(-0.1, 0.1)
                mod.running_mean.uniform_(-0.1, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        for p in list(m.rpn_head.parameters()) + list(m.neck.parameters()):
            if p.dim() == 1:
                p.uniform_(-0.1, 0.1)
    return m


def _trunk_run(m, img, train, enabled):
    from detmatch_amd import chain
    old = chain.ENABLED
    chain.ENABLED = enabled
    try:
        m.train(train)
        for p in m.parameters():
            p.grad = None
        with torch.set_grad_enabled(train):
            x, cls, reg, raw = m._trunk(img)
        out = dict(x=[t.detach().clone() for t in x], raw=[t.detach().clone() for t in raw])
        if train:
            g = torch.Generator(device='cpu').manual_seed(2)
            loss = sum((t * torch.randn(t.shape, generator=g).to(t.device)).sum() for t in list(x) + list(raw))
            loss.backward()
            out['grads'] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        return out
    finally:
        chain.ENABLED = old


@pytest.mark.parametrize('train', [True, False])
def test_frcnn_trunk_chain_equals_op_by_op(dev, train):
    m = _frcnn(dev)
    if not train:
        for p in m.parameters():
            p.requires_grad_(False)
    img = torch.randn(2, 3, 128, 192, device=dev) * 50
    ref = _trunk_run(m, img, train, enabled=False)
    got = _trunk_run(m, img, train, enabled=True)
    assert m.__dict__.get('_trunk_chains'), 'the chain did not run'
    ch = list(m.__dict__['_trunk_chains'].values())[0]
    assert ch.launches()[0] >= 70
    for i, (a, b) in enumerate(zip(got['x'] + got['raw'], ref['x'] + ref['raw'])):
        _same(a, b, 'trunk output %d' % i)
    if train:
        assert set(got['grads']) == set(ref['grads']) and len(ref['grads']) > 40
        for k, b in ref['grads'].items():
            # tensors with three gradient contributions (a pyramid input feeds the lateral conv and both branches of
            # the next stage) are summed in another order than autograd's: equal to fp32 rounding, not bit for bit
            a = got['grads'][k]
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7, \
                'grad %s: %g vs max %g' % (k, float((a - b).abs().max()), float(b.abs().max()))


# ---- the element-wise / layout entry points of csrc/chain_ops.hip against plain torch ---------------------------------
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_chain_ops_match_torch(dev):
    import torch.nn.functional as F
    from detmatch_amd import _lib
    L = _lib.lib()
    st = _lib.stream()
    g = torch.Generator(device='cpu').manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    # relu mask / add + mask
    a, b, y = rnd(5, 1028), rnd(5, 1028), rnd(5, 1028)
    out = torch.empty_like(a)
    _lib.check(L.dm_relu_mask_f32(a.data_ptr(), y.data_ptr(), out.data_ptr(), a.numel(), st), 'relu_mask')
    assert torch.equal(out, torch.where(y > 0, a, torch.zeros_like(a)))
    _lib.check(L.dm_add_mask_f32(a.data_ptr(), b.data_ptr(), y.data_ptr(), out.data_ptr(), a.numel(), st), 'add_mask')
    assert torch.equal(out, torch.where(y > 0, a + b, torch.zeros_like(a)))
    _lib.check(L.dm_add_mask_f32(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), a.numel(), st), 'add')
    assert torch.equal(out, a + b)
    # column sums (bias gradient): float64 reference, accumulate flag, run-to-run identical
    for rows, c in ((70400, 72 + 0), (333, 256), (31, 4), (122880, 256)):
        c = (c + 3) // 4 * 4
        x = rnd(rows, c)
        ws = torch.empty(L.dm_colsum_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
        o1 = torch.full((c,), 7.0, device=dev)
        _lib.check(L.dm_colsum_f32(x.data_ptr(), rows, c, o1.data_ptr(), 0, ws.data_ptr(), ws.numel(), st), 'colsum')
        want = x.double().sum(0)
        assert float((o1.double() - want).abs().max()) <= 2e-6 * float(x.abs().sum(0).max())
        o2 = o1.clone()
        _lib.check(L.dm_colsum_f32(x.data_ptr(), rows, c, o2.data_ptr(), 1, ws.data_ptr(), ws.numel(), st), 'colsum')
        o3 = torch.empty_like(o1)
        _lib.check(L.dm_colsum_f32(x.data_ptr(), rows, c, o3.data_ptr(), 0, ws.data_ptr(), ws.numel(), st), 'colsum')
        assert torch.equal(o3, o1) and torch.allclose(o2, 2 * o1, rtol=1e-6)
    # nearest resize and its gradient (exact 2x and a ragged size), max-pool, sub-sampling gradient
    for (hi, wi), (ho, wo) in (((12, 39), (24, 78)), ((7, 5), (16, 13)), ((24, 78), (12, 39))):
        x = rnd(2, 8, hi, wi)
        xn = _nhwc(x)
        yn = torch.empty(2, ho, wo, 8, device=dev)
        _lib.check(L.dm_resize_nearest_nhwc(xn.data_ptr(), 2, hi, wi, 8, ho, wo, yn.data_ptr(), st), 'resize')
        want = F.interpolate(x, size=(ho, wo), mode='nearest')
        assert torch.equal(yn, _nhwc(want))
        gy = rnd(2, 8, ho, wo)
        xr = x.clone().requires_grad_(True)
        F.interpolate(xr, size=(ho, wo), mode='nearest').backward(gy)
        gx = torch.full((2, hi, wi, 8), 3.0, device=dev)
        _lib.check(L.dm_resize_nearest_nhwc_backward(_nhwc(gy).data_ptr(), 2, hi, wi, 8, ho, wo, gx.data_ptr(), 0, st), 'rb')
        assert torch.allclose(gx, _nhwc(xr.grad), rtol=1e-6, atol=1e-6)
        _lib.check(L.dm_resize_nearest_nhwc_backward(_nhwc(gy).data_ptr(), 2, hi, wi, 8, ho, wo, gx.data_ptr(), 1, st), 'rb')
        assert torch.allclose(gx, 2 * _nhwc(xr.grad), rtol=1e-6, atol=1e-6)
    x = rnd(2, 8, 33, 47)
    x[0, 0, 3, 3] = float('nan')
    for k, s, p in ((3, 2, 1), (1, 2, 0), (2, 2, 0)):
        want = F.max_pool2d(x, k, s, p)
        yn = torch.empty(_nhwc(want).shape, device=dev)
        _lib.check(L.dm_maxpool_nhwc(_nhwc(x).data_ptr(), 2, 33, 47, 8, k, s, p, yn.data_ptr(), st), 'maxpool')
        assert torch.equal(torch.nan_to_num(yn, nan=123.0), torch.nan_to_num(_nhwc(want), nan=123.0))
    xr = rnd(2, 8, 33, 47).requires_grad_(True)
    yr = F.max_pool2d(xr, 1, 2)
    gy = rnd(*yr.shape)
    yr.backward(gy)
    gx = torch.empty(2, 33, 47, 8, device=dev)
    _lib.check(L.dm_subsample_nhwc_backward(_nhwc(gy).data_ptr(), 2, 33, 47, 8, 2, gx.data_ptr(), st), 'subsample')
    assert torch.equal(gx, _nhwc(xr.grad))
    # pitched copies: 16-byte path and the scalar path (18 | 42 | 12 column blocks)
    src = rnd(1000, 72)
    for off, wd in ((0, 18), (18, 42), (60, 12), (8, 64)):
        dst = torch.zeros(1000, wd, device=dev)
        _lib.check(L.dm_copy2d_f32(src.data_ptr() + off * 4, 72, dst.data_ptr(), wd, 1000, wd, st), 'copy2d')
        assert torch.equal(dst, src[:, off:off + wd])
    # multi-tensor add / assign
    import numpy as np
    from detmatch_amd.dense_chain import _AXPY_DESC
    ds, ss = [rnd(n) for n in (5, 1000, 77)], [rnd(n) for n in (5, 1000, 77)]
    want = [d + s_ for d, s_ in zip(ds, ss)]
    rows = np.array([(d.data_ptr(), s_.data_ptr(), d.numel()) for d, s_ in zip(ds, ss)], dtype=_AXPY_DESC)
    tab = torch.from_numpy(rows.view(np.uint8).copy()).to(dev)
    _lib.check(L.dm_multi_add_f32(tab.data_ptr(), 3, 1000, 0, st), 'multi_add')
    assert all(torch.equal(d, w) for d, w in zip(ds, want))
    _lib.check(L.dm_multi_add_f32(tab.data_ptr(), 3, 1000, 1, st), 'multi_assign')
    assert all(torch.equal(d, s_) for d, s_ in zip(ds, ss))


# ---- StackSAModuleMSG as one chained call (sa_chain.py) ----------------------------------------------------------------
def _sa_run(sa, args, feats, train, enabled, own_wgrad):
    from detmatch_amd import chain, pointnet2_stack as pn
    old, old_w = chain.ENABLED, pn.TallSkinnyLinear.OWN_WGRAD
    chain.ENABLED, pn.TallSkinnyLinear.OWN_WGRAD = enabled, own_wgrad
    try:
        sa.train(train)
        for p in sa.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(train)
        with torch.set_grad_enabled(train):
            _, out = sa(*args, features=f)
        res = dict(out=out.detach().clone())
        if train:
            g = torch.Generator(device='cpu').manual_seed(4)
            (out * torch.randn(out.shape, generator=g).to(out.device)).sum().backward()
            res['gf'] = f.grad.clone()
            res['grads'] = {n: p.grad.clone() for n, p in sa.named_parameters()}
        res['state'] = {k: v.clone() for k, v in sa.state_dict().items()}
        return res
    finally:
        chain.ENABLED, pn.TallSkinnyLinear.OWN_WGRAD = old, old_w


@pytest.mark.parametrize('train', [True, False])
def test_sa_module_chain_equals_op_by_op(dev, train):
    from detmatch_amd import pointnet2_stack as pn
    torch.manual_seed(5)
    g = torch.Generator(device='cpu').manual_seed(6)
    c, m_per, counts = 32, 1024, [5000, 3777]
    sa = pn.StackSAModuleMSG(radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[c, 32, 32], [c, 32, 64]], use_xyz=True,
                             pool_method='max_pool').to(dev)
    with torch.no_grad():
        for mod in sa.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5), mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.1, 0.1), mod.running_var.uniform_(0.5, 2.0)
    state = copy.deepcopy(sa.state_dict())
    n = sum(counts)
    xyz = (torch.rand(n, 3, generator=g) * torch.tensor([20.0, 20.0, 3.0])).to(dev)
    feats = torch.randn(n, c, generator=g).to(dev)
    new_xyz = torch.cat([xyz[:counts[0]][:m_per], xyz[counts[0]:][:m_per]]).contiguous() + 0.05
    args = (xyz, torch.tensor(counts, dtype=torch.int32, device=dev), new_xyz,
            torch.tensor([m_per, m_per], dtype=torch.int32, device=dev))
    ref = _sa_run(sa, args, feats, train, enabled=False, own_wgrad=True)
    sa.load_state_dict(state)
    got = _sa_run(sa, args, feats, train, enabled=True, own_wgrad=True)
    assert any(v is not False for v in sa.__dict__.get('_chains', {}).values()), 'the chain did not run'
    _same(got['out'], ref['out'], 'pooled features')
    for k in ref['state']:
        _same(got['state'][k], ref['state'][k], 'state ' + k)
    if train:
        # (the scatter of the grouped rows' gradient accumulates with atomic adds: equal to rounding, run to run)
        assert float((got['gf'] - ref['gf']).abs().max()) <= 1e-5 * float(ref['gf'].abs().max())
        for k in ref['grads']:
            _same(got['grads'][k], ref['grads'][k], 'grad ' + k)
        # against the op-by-op default (weight gradients of the tall-skinny GEMMs on a batched BLAS call): fp32 rounding
        sa.load_state_dict(state)
        blas = _sa_run(sa, args, feats, train, enabled=False, own_wgrad=False)
        for k in blas['grads']:
            a, b = got['grads'][k], blas['grads'][k]
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, k


# ---- VoxelBackBone8x as one chained call (sparse_chain.py) ---------------------------------------------------------------
def _backbone_run(bb, vf, coords, train, enabled):
    from detmatch_amd import chain
    from detmatch_amd.spconv.ops import deferred_weight_grads
    old = chain.ENABLED
    chain.ENABLED = enabled
    try:
        bb.train(train)
        for p in bb.parameters():
            p.grad = None
        with torch.set_grad_enabled(train):
            d = bb(dict(voxel_features=vf, voxel_coords=coords, batch_size=2))
        feats = [d['multi_scale_3d_features'][k] for k in ('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4')] + \
            [d['encoded_spconv_tensor']]
        res = dict(feats=[t.features.detach().clone() for t in feats], idx=[t.indices.clone() for t in feats],
                   shapes=[list(t.spatial_shape) for t in feats])
        if train:
            g = torch.Generator(device='cpu').manual_seed(8)
            loss = sum((t.features * torch.randn(t.features.shape, generator=g).to(vf.device)).sum() for t in feats)
            with deferred_weight_grads():      # the batched weight-gradient launch of the op-by-op path
                loss.backward()
            res['grads'] = {n: p.grad.clone() for n, p in bb.named_parameters()}
        res['state'] = {k: v.clone() for k, v in bb.state_dict().items()}
        return res
    finally:
        chain.ENABLED = old


@pytest.mark.parametrize('train', [True, False])
def test_voxel_backbone_chain_equals_op_by_op(dev, train):
    from detmatch_amd import synth, voxel
    from detmatch_amd.pcdet.backbones_3d import VoxelBackBone8x
    torch.manual_seed(1)
    bb = VoxelBackBone8x({}, 4, [1408, 1600, 40]).to(dev)
    with torch.no_grad():
        for mod in bb.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.5, 1.5), mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.1, 0.1), mod.running_var.uniform_(0.5, 2.0)
    state = copy.deepcopy(bb.state_dict())
    pts = [torch.from_numpy(synth.lidar_frame(i)['points']).to(dev) for i in range(2)]
    _, coords, _, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    ref = _backbone_run(bb, mean, coords, train, enabled=False)
    bb.load_state_dict(state)
    got = _backbone_run(bb, mean, coords, train, enabled=True)
    assert bb.__dict__.get('_chains', {}).get(train) not in (None, False), 'the chain did not run'
    assert got['shapes'] == ref['shapes']
    for i, (a, b) in enumerate(zip(got['idx'], ref['idx'])):
        assert torch.equal(a, b), 'indices of level %d' % i
    for i, (a, b) in enumerate(zip(got['feats'], ref['feats'])):
        _same(a, b, 'features of level %d' % i)
    for k in ref['state']:
        _same(got['state'][k], ref['state'][k], 'state ' + k)
    if train:
        for k in ref['grads']:
            _same(got['grads'][k], ref['grads'][k], 'grad ' + k)


def test_sa_module_chain_on_padded_raw_point_features(dev):
    """The raw-points source of VoxelSetAbstraction has ONE feature column (intensity): the chain pads it to four (zero
    columns meeting zero weight columns) so that the grouped rows are 16-byte aligned; the op-by-op path runs its 5-wide
    first layer on BLAS — equal to fp32 rounding, parameters' gradients included; the features carry no gradient."""
    from detmatch_amd import pointnet2_stack as pn
    torch.manual_seed(9)
    g = torch.Generator(device='cpu').manual_seed(10)
    m_per, counts = 1024, [6000, 5000]
    sa = pn.StackSAModuleMSG(radii=[0.4, 0.8], nsamples=[16, 16], mlps=[[1, 16, 16], [1, 16, 16]], use_xyz=True,
                             pool_method='max_pool').to(dev)
    state = copy.deepcopy(sa.state_dict())
    n = sum(counts)
    xyz = (torch.rand(n, 3, generator=g) * torch.tensor([10.0, 10.0, 2.0])).to(dev)
    feats = torch.rand(n, 1, generator=g).to(dev)
    new_xyz = torch.cat([xyz[:counts[0]][:m_per], xyz[counts[0]:][:m_per]]).contiguous() + 0.02
    args = (xyz, torch.tensor(counts, dtype=torch.int32, device=dev), new_xyz,
            torch.tensor([m_per, m_per], dtype=torch.int32, device=dev))

    def run(enabled):
        from detmatch_amd import chain
        old = chain.ENABLED
        chain.ENABLED = enabled
        try:
            sa.load_state_dict(state)
            sa.train()
            for p in sa.parameters():
                p.grad = None
            _, out = sa(*args, features=feats)
            gg = torch.Generator(device='cpu').manual_seed(11)
            (out * torch.randn(out.shape, generator=gg).to(dev)).sum().backward()
            return out.detach().clone(), {k: p.grad.clone() for k, p in sa.named_parameters()}
        finally:
            chain.ENABLED = old
    ref_out, ref_g = run(False)
    got_out, got_g = run(True)
    assert any(v is not False for v in sa.__dict__.get('_chains', {}).values()), 'the chain did not run'
    assert float((got_out - ref_out).abs().max()) <= 1e-4 * float(ref_out.abs().max())
    for k in ref_g:
        assert float((got_g[k] - ref_g[k]).abs().max()) <= 1e-3 * float(ref_g[k].abs().max()) + 1e-6, k
