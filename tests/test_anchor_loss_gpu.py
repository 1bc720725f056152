"""Fused anchor-head loss kernel (csrc/anchor_loss.hip) against the element-wise restatement of the
reference's anchor_head_template.py:101-214 (the one pinned by the reference-generated goldens in
tests/test_pcdet_torch_golden.py): the three losses and all three gradients, full KITTI anchor grid."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs, synth
from detmatch_amd.pcdet.config import ConfigDict
from detmatch_amd.pcdet.dense_heads import AnchorHeadSingle

pytestmark = pytest.mark.gpu


def _head(dev, c_in=16):
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    h = AnchorHeadSingle(cfg.DENSE_HEAD, input_channels=c_in, num_class=3, class_names=configs.CLASS_NAMES,
                         grid_size=np.array([1408, 1600, 40]),
                         point_cloud_range=np.array(configs.POINT_CLOUD_RANGE, dtype=np.float32)).to(dev)
    with torch.no_grad():
        for p in h.parameters():
            p.add_(torch.randn_like(p) * 0.2)
    return h.train()


def _gt(n_samples):
    gt = np.zeros((n_samples, 12, 8), np.float32)
    for s in range(n_samples):
        f = synth.lidar_frame(s)
        lab = synth._SIM_TO_CFG_LABEL[f['gt_labels']] + 1
        g = np.concatenate([f['gt_boxes'], lab[:, None].astype(np.float32)], 1)
        if s == n_samples - 1:
            g = g[:0]                      # a sample without any GT: normaliser clamps to 1
        gt[s, :len(g)] = g
    return gt


def test_fused_losses_and_gradients_match_the_elementwise_path():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    h = _head(dev)
    x = torch.randn(3, 16, 200, 176, device=dev, requires_grad=True)
    gt = torch.from_numpy(_gt(3)).to(dev)
    res = {}
    for mode in ('torch', 'fused'):
        x.grad = None
        h.zero_grad()
        h(dict(spatial_features_2d=x, gt_boxes=gt, batch_size=3))
        assert int((h.forward_ret_dict['box_cls_labels'] > 0).sum()) > 50
        assert int((h.forward_ret_dict['box_cls_labels'] < 0).sum()) > 0
        loss, tb = h.get_loss_torch() if mode == 'torch' else h.get_loss()
        (loss * 1.7).backward()            # a non-unit upstream gradient
        res[mode] = (loss.detach(), {k: v.clone() for k, v in tb.items()}, x.grad.clone(),
                     h.conv_cls.weight.grad.clone(), h.conv_box.weight.grad.clone(), h.conv_dir_cls.weight.grad.clone())
    (lt, tbt, gxt, gct, gbt, gdt), (lf, tbf, gxf, gcf, gbf, gdf) = res['torch'], res['fused']
    assert torch.allclose(lf, lt, rtol=1e-5), (lf, lt)
    for k in ('rpn_loss_cls', 'rpn_loss_loc', 'rpn_loss_dir', 'rpn_loss'):
        assert torch.allclose(tbf[k], tbt[k], rtol=1e-5, atol=1e-7), (k, tbf[k], tbt[k])
    for a, b, name in ((gxf, gxt, 'x'), (gcf, gct, 'cls'), (gbf, gbt, 'box'), (gdf, gdt, 'dir')):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) <= 2e-5 * scale + 1e-9, (name, float((a - b).abs().max()), scale)


def test_fused_loss_is_deterministic_and_handles_nan_targets():
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    h = _head(dev)
    x = torch.randn(2, 16, 200, 176, device=dev)
    h(dict(spatial_features_2d=x, gt_boxes=torch.from_numpy(_gt(2)).to(dev), batch_size=2))
    a, _ = h.get_loss()
    b, _ = h.get_loss()
    assert torch.equal(a, b)                                   # fixed-order reduction
    # NaN regression targets on positive anchors are neutralised (loss_utils.py:117), as in the torch path
    d = h.forward_ret_dict
    pos = (d['box_cls_labels'] > 0).nonzero()
    d['box_reg_targets'] = d['box_reg_targets'].clone()
    d['box_reg_targets'][pos[0, 0], pos[0, 1], 2] = float('nan')
    lf, _ = h.get_loss()
    lt, _ = h.get_loss_torch()
    assert torch.isfinite(lf) and torch.allclose(lf, lt, rtol=1e-5)
