"""Fused cost matrix of the 2D <-> 3D Hungarian matching (csrc/ssl_match.hip) against the tensor chain
of FusionHungarianMatching.match (pinned to the reference's assigner by tests/test_ssl_host.py goldens):
same matched index pairs, matched costs to fp32 accuracy, with and without the 3D -> 2D projection."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs

pytestmark = pytest.mark.gpu


def _module(project):
    from detmatch_amd.mm3d.ssl_modules import FusionHungarianMatching
    return FusionHungarianMatching(assigner_cfg=configs._hung_assigner(), cost_thr=-1.5, img_metas='m',
                                   cls_includes_bg_pred_3d=False, cls_includes_bg_pred_2d=True,
                                   in_bboxes_3d_key='a', in_bboxes_2d_key='b', out_bboxes_3d_key='c',
                                   out_bboxes_2d_key='d', project_3d_to_2d=project)


def _scene(dev, seed, n3, n2):
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    g = torch.Generator().manual_seed(seed)
    b3 = torch.zeros(n3, 7)
    b3[:, 0] = torch.rand(n3, generator=g) * 50 + 5            # in front of the camera
    b3[:, 1] = (torch.rand(n3, generator=g) - 0.5) * 30
    b3[:, 2] = -1.7 + 0.3 * torch.rand(n3, generator=g)
    b3[:, 3:6] = torch.tensor([3.9, 1.6, 1.56]) * (0.7 + 0.6 * torch.rand(n3, 3, generator=g))
    b3[:, 6] = (torch.rand(n3, generator=g) - 0.5) * 6.28
    # KITTI-like projection: camera looks along +x of the LiDAR frame
    P = np.array([[721.5, 0, 609.5, 44.9], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003], [0, 0, 0, 1]], np.float32)
    Tr = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], np.float32)
    l2i = P @ Tr
    meta = dict(lidar2img=l2i, ori_shape=(375, 1242, 3))
    s3 = torch.rand(n3, 3, generator=g) * torch.rand(n3, 1, generator=g)
    c = torch.rand(n2, 2, generator=g) * torch.tensor([1242.0, 375.0])
    wh = torch.rand(n2, 2, generator=g) * torch.tensor([300.0, 150.0]) + 5
    b2 = torch.cat([(c - wh / 2).clamp(min=0), torch.minimum(c + wh / 2, torch.tensor([1242.0, 375.0]))], 1)
    s2 = torch.rand(n2, 4, generator=g)
    s2 = s2 / s2.sum(1, keepdim=True)
    return LiDARInstance3DBoxes(b3.to(dev)), s3.to(dev), b2.to(dev), s2.to(dev), meta


@pytest.mark.parametrize('n3,n2,project', [(37, 21, True), (5, 60, True), (64, 64, False), (1, 1, True), (120, 7, False)])
def test_fused_match_equals_tensor_chain(dev, n3, n2, project):
    from detmatch_amd.mm3d.bbox_utils import bbox_3d_to_bbox_2d
    m = _module(project)
    m.cost_thr = 10.0 if n3 * n2 > 1 else None                 # keep (nearly) all pairs: compares more
    boxes3, s3, b2, s2, meta = _scene(dev, n3 * 100 + n2, n3, n2)
    if project:
        e3 = (boxes3, s3)
    else:
        e3 = (bbox_3d_to_bbox_2d(boxes3, meta['lidar2img'], meta['ori_shape'])[0].requires_grad_(True), s3)
    e2 = (b2, s2)
    i3, i2, c = m.match(e3, e2, meta)
    j3, j2, d = m.match(e3, e2, meta, fused=False)
    assert i3.dtype == torch.int64 and torch.equal(i3, j3) and torch.equal(i2, j2)
    np.testing.assert_allclose(c.cpu().numpy(), d.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert len(i3) == min(n3, n2) or m.cost_thr is not None
    # the module's outputs are gathers of its inputs: gradients still reach the projected student boxes
    out = {}
    from detmatch_amd.mm3d.ssl_modules import mlvl_set
    bd = dict(a=[e3], b=[e2], m=[meta])
    bd = m.forward(None, bd)
    if not project:
        assert bd['c'][0][0].requires_grad
    m.cost_thr = -1.5
    k3, k2, _ = m.match(e3, e2, meta)
    l3, l2, _ = m.match(e3, e2, meta, fused=False)
    assert torch.equal(k3, l3) and torch.equal(k2, l2)
