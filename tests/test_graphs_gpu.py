"""graphs.StaticSection: a captured inference section replays the CURRENT weights and statistics (packs and
folded-BatchNorm maps are recorded inside the graph), and equals the plain call bit for bit."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def test_static_section_equals_plain_call_and_sees_weight_updates(dev, monkeypatch):
    from detmatch_amd import dense_conv, graphs
    from detmatch_amd.mm2d.backbone import FrozenBN, conv_frozen_bn
    torch.manual_seed(0)
    c1 = nn.Conv2d(32, 64, 3, padding=1, bias=False).to(dev)
    b1 = FrozenBN(64).to(dev)
    c2 = nn.Conv2d(64, 32, 1, bias=True).to(dev)
    with torch.no_grad():
        b1.running_var.uniform_(0.5, 2.0)
        b1.running_mean.normal_()

    def fn(x):
        y = conv_frozen_bn(x, c1, b1, relu=True)
        return dense_conv.conv2d(y, c2.weight, c2.bias, 1, 0), y

    sec = graphs.StaticSection(fn, 'test')
    monkeypatch.setattr(graphs, 'ENABLED', True)       # opt-in in the product (DM_HIPGRAPH=1)
    with torch.no_grad():
        for it in range(7):
            x = torch.randn(2, 32, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
            if it == 4:            # in-place update (version counter moves)
                c1.weight.mul_(1.25)
                b1.running_mean.add_(0.5)
            if it == 5:            # update through raw pointers, as the fused optimizer / EMA kernels do
                c2.weight.data.view(-1)[::3] *= -1.0
                dense_conv.weights_changed()
            got = sec(x)
            got = [g.clone() for g in got]
            want = fn(x)
            for g, w in zip(got, want):
                assert torch.equal(g, w), it
    assert sec.captures == 1 and sec.replays == 7 - graphs._WARMUP_CALLS
    # with autograd on, the section is a plain call
    x = torch.randn(2, 32, 24, 40, device=dev, requires_grad=True)
    out = sec(x)[0]
    out.sum().backward()
    assert x.grad is not None and sec.replays == 7 - graphs._WARMUP_CALLS


def _chain(dev):
    """conv -> train-mode BatchNorm + ReLU -> conv(+bias): what a BEV block is made of."""
    from detmatch_amd import dense_conv
    from detmatch_amd.pcdet.backbones_2d import _bn_relu_nhwc
    torch.manual_seed(0)
    c1 = nn.Conv2d(16, 32, 3, padding=1, bias=False).to(dev)
    bn = nn.BatchNorm2d(32, eps=1e-3, momentum=0.01).to(dev)
    c2 = nn.Conv2d(32, 8, 1, bias=True).to(dev)

    def fn(x):
        y = _bn_relu_nhwc(dense_conv.conv2d(x, c1.weight, None, 1, 1), bn)
        z = dense_conv.conv2d(y, c2.weight, c2.bias, 1, 0)
        return z, y.detach()
    return fn, (c1, bn, c2)


def test_train_section_equals_plain_call(dev, monkeypatch):
    """Forward values, input / parameter gradients, BatchNorm running statistics and call counters of a
    TrainSection equal the plain call's over several iterations with weight updates in between (in place and
    through raw pointers), two calls in flight per iteration (two instances), backward in reverse order."""
    import copy
    from detmatch_amd import dense_conv, graphs
    monkeypatch.setattr(graphs, 'TRAIN_ENABLED', True)
    fn_a, mods_a = _chain(dev)
    fn_b, mods_b = _chain(dev)
    for ma, mb in zip(mods_a, mods_b):
        mb.load_state_dict(copy.deepcopy(ma.state_dict()))
    pa = [p for m in mods_a for p in m.parameters()]
    pb = [p for m in mods_b for p in m.parameters()]
    sec = graphs.TrainSection(fn_a, mods_a, 'test.train')
    keep_alive = []         # graphs of earlier plain calls stay alive: the parameters' AccumulateGrad nodes (made on
    #                         the caller's stream) must not leak that stream into the backward capture
    g = torch.Generator(device='cpu').manual_seed(1)
    for it in range(8):
        graphs.new_iteration()
        xs = [torch.randn(2, 16, 20, 24, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
              for _ in range(2)]
        if it == 5:
            with torch.no_grad():
                for p, q in zip(pa, pb):
                    p.mul_(1.1), q.mul_(1.1)
        if it == 6:
            with torch.no_grad():
                for p, q in zip(pa, pb):
                    p.data.view(-1)[::2] *= -1.0
                    q.data.view(-1)[::2] *= -1.0
            dense_conv.weights_changed()
        res = []
        for fn, params, graphed in ((sec, pa, True), (fn_b, pb, False)):
            for p in params:
                p.grad = None
            ins = [x.clone().requires_grad_(True) for x in xs]
            outs = [fn(i) for i in ins]
            keep_alive.append(outs)
            vals = [(o[0].detach().clone(), o[1].clone()) for o in outs]
            w = [torch.randn(o[0].shape, generator=g).to(dev) for o in outs] if graphed else res[0][3]
            (outs[1][0] * w[1]).sum().backward()        # the second call's backward first
            (outs[0][0] * w[0]).sum().backward()
            res.append((vals, [i.grad.clone() for i in ins], [p.grad.clone() for p in params], w))
        for (za, ya), (zb, yb) in zip(res[0][0], res[1][0]):
            assert torch.equal(za, zb) and torch.equal(ya, yb), it
        for ga, gb in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
            torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-6)
        for ba, bb in zip([b for m in mods_a for b in m.buffers()], [b for m in mods_b for b in m.buffers()]):
            assert torch.equal(ba, bb), it
    assert sec.captures == 2 and sec.replays >= 8 and sec.bwd_replays == sec.replays
