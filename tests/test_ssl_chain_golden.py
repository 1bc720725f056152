"""The pseudo-label chain modules against goldens produced by the REFERENCE's own modules
(tests/golden/gen_ssl_chain_golden.py: Bboxes3DTo2D, FusionHungarianMatching, HungarianConsistency of
mmdet3d/models/ssl_modules/** run on seeded inputs, gradients from the reference's autograd):

  * on the CPU the host (tensor) formulation is checked,
  * on the GPU (-m gpu) the SAME inputs go straight into the fused kernels — csrc/box_project.hip
    (dm_unaug_project fwd / bwd), csrc/ssl_match.hip (dm_fusion_match_cost + dm_lap_host),
    csrc/consistency_loss.hip — and are compared with the stored reference outputs: validity masks, kept
    sets and Hungarian assignments bit-exact, boxes / costs / losses 1e-3 relative or tighter.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from detmatch_amd import configs

G = np.load(os.path.join(GOLDEN, 'ssl_chain.npz'))
DEVICES = ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)]


def _device(name):
    if name == 'cuda' and not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device(name if name == 'cpu' else 'cuda:0')


def _bare_ssl():
    from detmatch_amd.mm3d.ssl import SSL
    s = SSL.__new__(SSL)
    torch.nn.Module.__init__(s)
    return s


def _assert_fused(device, on):
    """On the GPU the module must have taken the kernel, not the tensor chain."""
    from detmatch_amd import fused
    if device.type == 'cuda':
        assert fused.ENABLED and on


@pytest.mark.parametrize('devname', DEVICES)
@pytest.mark.parametrize('case', range(int(G['n_proj'])))
def test_bboxes3d_to_2d_matches_reference_module(devname, case):
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    from detmatch_amd.mm3d.ssl_modules import Bboxes3DTo2D
    dev = _device(devname)
    k = 'pj%d_' % case
    flow = [f for f in str(G[k + 'flow']).split(',') if f]
    meta = dict(pcd_rotation=torch.from_numpy(G[k + 'rotation']), pcd_scale_factor=float(G[k + 'scale']),
                pcd_trans=G[k + 'trans'], pcd_horizontal_flip=bool(G[k + 'hflip']),
                pcd_vertical_flip=bool(G[k + 'vflip']), transformation_3d_flow=flow,
                lidar2img=G['lidar2img'], ori_shape=(375, 1242, 3), img_shape=(384, 1248, 3))
    t = torch.from_numpy(G[k + 'boxes']).to(dev).requires_grad_(True)
    n = t.shape[0]
    ids = torch.arange(n, device=dev)
    scores = torch.rand(n, 3, device=dev)
    mod = Bboxes3DTo2D(img_metas='stu.img_metas', in_bboxes_key='stu.in', out_bboxes_key='stu.out', filter_invalid=False)
    boxes = LiDARInstance3DBoxes(t.detach())
    boxes.tensor = t                                               # keep the graph (the reference's ctor clones)
    bd = dict(stu=dict(img_metas=[meta]))
    bd['stu']['in'] = [(boxes, scores, ids)]
    xy = mod.forward(None, bd)['stu']['out'][0][0]
    if dev.type == 'cuda':
        assert 'UnaugProject' in type(xy.grad_fn).__name__          # the kernel, not the tensor chain
    want = G[k + 'xyxy']
    valid = G[k + 'valid']
    got = xy.detach().cpu().numpy()
    # pixel coordinates up to 1242: 2e-2 px absolute = 2e-5 of the range (clamped-depth corners amplify)
    np.testing.assert_allclose(got[valid], want[valid], rtol=1e-4, atol=2e-2)
    (xy * torch.from_numpy(G[k + 'up']).to(dev)).sum().backward()
    gw = G[k + 'grad']
    gg = t.grad.cpu().numpy()
    scale = np.abs(gw[valid]).max()
    assert np.abs(gg[valid] - gw[valid]).max() <= 2e-3 * scale
    # kept set of filter_invalid=True: exact
    modf = Bboxes3DTo2D(img_metas='stu.img_metas', in_bboxes_key='stu.in', out_bboxes_key='stu.out', filter_invalid=True)
    bd = dict(stu=dict(img_metas=[meta]))
    bd['stu']['in'] = [(LiDARInstance3DBoxes(t.detach()), scores, ids)]
    from detmatch_amd.mm3d.ssl_modules import plain       # (on the GPU the filter stays a mask until somebody needs the list)
    kept = plain(modf.forward(None, bd)['stu']['out'])[0][2].cpu().numpy()
    assert np.array_equal(kept, G[k + 'kept_ids'])


@pytest.mark.parametrize('devname', DEVICES)
@pytest.mark.parametrize('case', range(int(G['n_match'])))
def test_fusion_hungarian_matching_matches_reference_module(devname, case):
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    from detmatch_amd.mm3d.ssl_modules import FusionHungarianMatching
    dev = _device(devname)
    k = 'fm%d_' % case
    project = bool(G[k + 'project'])
    thr = float(G[k + 'cost_thr'])
    mod = FusionHungarianMatching(assigner_cfg=configs._hung_assigner(), cost_thr=None if np.isnan(thr) else thr,
                                  img_metas='m', cls_includes_bg_pred_3d=False, cls_includes_bg_pred_2d=True,
                                  in_bboxes_3d_key='a', in_bboxes_2d_key='b', out_bboxes_3d_key='c',
                                  out_bboxes_2d_key='d', match_cost_key='e', project_3d_to_2d=project)
    n3, n2 = len(G[k + 'scores3d']), len(G[k + 'scores2d'])
    first = (LiDARInstance3DBoxes(torch.from_numpy(G[k + 'boxes3d']).to(dev)) if project
             else torch.from_numpy(G[k + 'proj']).to(dev))
    e3 = (first, torch.from_numpy(G[k + 'scores3d']).to(dev), torch.arange(n3, device=dev))
    e2 = (torch.from_numpy(G[k + 'boxes2d']).to(dev), torch.from_numpy(G[k + 'scores2d']).to(dev),
          torch.arange(n2, device=dev))
    meta = dict(lidar2img=G['lidar2img'], ori_shape=(375, 1242, 3))
    if dev.type == 'cuda' and n3 and n2:
        assert mod._device_costs() is not None                      # the fused cost kernel is eligible
    bd = mod.forward(None, dict(a=[e3], b=[e2], m=[meta]))
    i3, i2 = bd['c'][0][2].cpu().numpy(), bd['d'][0][2].cpu().numpy()
    assert np.array_equal(i3, G[k + 'idx3']) and np.array_equal(i2, G[k + 'idx2'])     # assignments: exact
    np.testing.assert_allclose(bd['e'][0].detach().cpu().numpy(), G[k + 'cost'], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('devname', DEVICES)
@pytest.mark.parametrize('case', range(int(G['n_cons'])))
def test_hungarian_consistency_matches_reference_module(devname, case):
    from detmatch_amd.mm3d.ssl_modules import HungarianConsistency
    dev = _device(devname)
    k = 'hc%d_' % case
    mod = HungarianConsistency(loss_cls_cfg=dict(type='FocalLoss', loss_weight=1.0, reduction='mean'),
                               loss_l1_cfg=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                               loss_iou_cfg=dict(type='GIoULoss', loss_weight=1.0, reduction='mean'),
                               loss_weights_cfg=dict(cls_loss=2, l1_loss=5 * 4, iou_loss=2),
                               cls_includes_bg_pred_in=False, cls_includes_bg_pred_target=True,
                               in_bboxes_key='stu.in', target_bboxes_key='tea.tgt',
                               target_img_metas_key='stu.img_metas', name='2D_to_3D_hung')
    ins, tgts = [], []
    for j in range(2):
        b = torch.from_numpy(G[k + 'pred%d' % j]).to(dev).requires_grad_(True)
        s = torch.from_numpy(G[k + 'ps%d' % j]).to(dev).requires_grad_(True)
        ins.append((b, s))
        tgts.append((torch.from_numpy(G[k + 'tgt%d' % j]).to(dev), torch.from_numpy(G[k + 'ts%d' % j]).to(dev)))
    bd = dict(stu=dict(img_metas=[dict(img_shape=(384, 1248, 3))] * 2), tea=dict(tgt=tgts), ssl_losses=dict())
    bd['stu']['in'] = ins
    if dev.type == 'cuda':
        assert mod._fusable(ins[0][0])
    losses = mod.forward(_bare_ssl(), bd)['ssl_losses']
    names = [n[len(k) + 5:] for n in G.files if n.startswith(k + 'loss_')]
    assert sorted(losses) == sorted(names)
    for name in names:
        assert float(losses[name].detach()) == pytest.approx(float(G[k + 'loss_' + name]), rel=1e-4, abs=1e-7), name
    sum(losses.values()).backward()
    for j, (b, s) in enumerate(ins):
        for got, want in ((b.grad, G[k + 'gpred%d' % j]), (s.grad, G[k + 'gps%d' % j])):
            if got is None:
                assert not np.any(want)
                continue
            scale = np.abs(want).max() + 1e-12
            assert np.abs(got.cpu().numpy() - want).max() <= 1e-3 * scale
