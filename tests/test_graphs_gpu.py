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
