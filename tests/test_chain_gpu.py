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
