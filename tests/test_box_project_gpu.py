"""csrc/box_project.hip (Bboxes3DTo2D on one sample in one launch, forward and backward) against the tensor
chain it replaces: apply_3d_transformation_bboxes(reverse=True) + bbox_3d_to_bbox_2d — both pinned to
reference goldens on the CPU (tests/golden/ssl_geometry.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _meta(rng, flow, hflip, vflip):
    ang = rng.uniform(-0.78, 0.78)
    c, s = np.cos(ang), np.sin(ang)
    # KITTI-like projection: camera looks along +x of the lidar frame
    p = np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2], [0, 0, 1, 0.003], [0, 0, 0, 1]], np.float32)
    r = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], np.float32)
    return dict(pcd_rotation=torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float32),
                pcd_scale_factor=float(rng.uniform(0.95, 1.05)),
                pcd_trans=np.asarray(rng.normal(0, 0.2, 3), np.float32),
                pcd_horizontal_flip=hflip, pcd_vertical_flip=vflip, transformation_3d_flow=flow,
                lidar2img=p @ r, ori_shape=(375, 1242, 3))


@pytest.mark.parametrize('flow,hflip,vflip', [(['HF', 'R', 'S', 'T'], True, False), (['R', 'S', 'T'], False, False),
                                              (['HF', 'VF', 'R', 'S', 'T'], True, True), ([], False, False),
                                              (['T', 'S', 'R', 'VF', 'HF'], False, True)])
def test_unaug_project_kernel_matches_tensor_chain(dev, flow, hflip, vflip):
    from detmatch_amd.mm3d.bbox_utils import (apply_3d_transformation_bboxes, bbox_3d_to_bbox_2d,
                                              unaug_project_boxes)
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    rng = np.random.default_rng(len(flow) * 7 + hflip + 2 * vflip)
    meta = _meta(rng, flow, hflip, vflip)
    n = 150
    b = np.concatenate([rng.uniform([2, -25, -2.5], [60, 25, 0.5], (n, 3)), rng.uniform([0.5, 0.4, 1.0], [4.5, 2.0, 2.2], (n, 3)),
                        rng.uniform(-3.2, 3.2, (n, 1))], 1).astype(np.float32)
    b[:10, 0] = rng.uniform(-20, 1.0, 10)              # behind / beside the camera: clamped depth, clipped boxes
    # the scene is laid out in the ORIGINAL frame; the student sees it through the recorded augmentations
    b = apply_3d_transformation_bboxes(LiDARInstance3DBoxes(torch.from_numpy(b).to(dev)), meta,
                                       reverse=False).tensor.cpu().numpy()
    g = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)).to(dev)
    res = []
    for fused in (True, False):
        t = torch.from_numpy(b).to(dev).requires_grad_(True)
        boxes = LiDARInstance3DBoxes(t * 1.0)
        if fused:
            xy, valid = unaug_project_boxes(boxes, meta)
        else:
            xy, valid = bbox_3d_to_bbox_2d(apply_3d_transformation_bboxes(boxes, meta, reverse=True),
                                           meta['lidar2img'], meta['ori_shape'])
        (xy * g).sum().backward()
        res.append((xy.detach(), valid, t.grad.clone()))
    (xa, va, ga), (xb, vb, gb) = res
    assert float((xa - xb).abs().max()) < 2e-2                      # pixels, coordinates up to 1242
    assert int((va != vb).sum()) <= 1 and int(va.sum()) > 50        # a corner exactly on the border may flip
    ok = (va & vb)
    scale = float(gb[ok].abs().max())
    assert float((ga[ok] - gb[ok]).abs().max()) < 2e-3 * scale
    # boxes clipped on all four sides give no gradient on either path
    dead = (xb[:, 0] == xb[:, 2]) | (xb[:, 1] == xb[:, 3])
    if bool(dead.any()):
        assert float(ga[dead].abs().max()) <= 1e-3 * scale + float(gb[dead].abs().max())


@pytest.mark.parametrize('n', [1, 7, 300])
def test_consistency_loss_kernel_matches_tensor_losses(dev, n):
    """csrc/consistency_loss.hip == mmdet-style FocalLoss(logit) + L1Loss(normalised) + GIoULoss of
    mm3d/losses.py (themselves pinned to reference goldens): values and gradients."""
    from detmatch_amd.mm3d.losses import FocalLoss, GIoULoss, L1Loss
    from detmatch_amd.mm3d.ssl_modules import _FusedConsistencyLoss
    g = torch.Generator().manual_seed(n)
    c = torch.rand(n, 2, generator=g) * torch.tensor([1100.0, 300.0]) + 60
    wh = torch.rand(n, 2, generator=g) * 150 + 5
    tgt = torch.cat([c - wh / 2, c + wh / 2], 1)
    pred = tgt + torch.randn(n, 4, generator=g) * 25
    pred[::5] = tgt[::5] + 400.0                                  # disjoint pairs
    if n > 2:
        pred[1, 2] = pred[1, 0] - 3.0                             # inverted box: negative width
        pred[2] = tgt[2]                                          # identical boxes: ties in min / max
    ps = torch.rand(n, 3, generator=g).clamp(1e-4, 1 - 1e-4)
    ps[0, 0] = 1e-9                                               # outside the logit clamp: no gradient
    ts = torch.rand(n, 3, generator=g)
    up = torch.tensor([2.0, 20.0, 2.0]).to(dev)
    res = []
    for fused in (True, False):
        b = pred.to(dev).requires_grad_(True)
        s = ps.to(dev).requires_grad_(True)
        if fused:
            vals = _FusedConsistencyLoss.apply(b, s, tgt.to(dev), ts.to(dev), 1248.0, 384.0, 0.25, 2.0)
        else:
            f = torch.tensor([1248.0, 384.0, 1248.0, 384.0]).to(dev)
            vals = torch.stack([FocalLoss()(torch.logit(s, eps=1e-6), torch.argmax(ts.to(dev), dim=1)),
                                L1Loss()(b / f, tgt.to(dev) / f), GIoULoss()(b, tgt.to(dev))])
        (vals * up).sum().backward()
        res.append((vals.detach(), b.grad.clone(), s.grad.clone()))
    (va, ba, sa), (vb, bb, sb) = res
    assert torch.allclose(va, vb, rtol=2e-5, atol=1e-6), (va, vb)
    assert float((ba - bb).abs().max()) <= 2e-4 * float(bb.abs().max()) + 1e-9
    assert float((sa - sb).abs().max()) <= 2e-4 * float(sb.abs().max()) + 1e-9
    assert float(sa[0, 0]) == 0.0 == float(sb[0, 0])


@pytest.mark.parametrize('flip', [False, True])
@pytest.mark.parametrize('ori2new', [False, True])
def test_bbox2d_transform_kernel_is_the_tensor_chain(dev, flip, ori2new):
    """dm_bbox2d_transform == bbox_2d_transform's tensor chain (pinned to the reference on the CPU): boxes
    bit for bit, gradient equal."""
    from detmatch_amd import fused
    from detmatch_amd.mm3d.bbox_utils import bbox_2d_transform
    meta = dict(img_shape=(384, 1248, 3), ori_shape=(375, 1242, 3), flip=flip,
                scale_factor=np.array([1.0048, 1.024, 1.0048, 1.024], np.float32),
                img_crop_offset=np.array([3.0, -2.0], np.float32))
    g = torch.Generator().manual_seed(3)
    b = torch.rand(40, 4, generator=g) * 1200
    up = torch.randn(40, 4, generator=g).to(dev)
    res = []
    prev = fused.ENABLED
    try:
        for on in (True, False):
            fused.ENABLED = on
            t = b.to(dev).requires_grad_(True)
            y = bbox_2d_transform(meta, t, ori2new)
            (y * up).sum().backward()
            res.append((y.detach(), t.grad.clone()))
    finally:
        fused.ENABLED = prev
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-6, atol=0)
