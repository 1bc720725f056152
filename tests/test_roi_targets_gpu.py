"""Second-stage target / loss kernels (csrc/roi_targets.hip) on the MI355X.

* the RoI-head and point-head losses against the REFERENCE's values (tests/golden/pcdet_roi_head.npz,
  pcdet_dense.npz: outputs of the reference's RoIHeadTemplate / PointHeadSimple), and their gradients
  against autograd through the tensor formulation (itself pinned to the same goldens on the CPU);
* the proposal-target kernel against the tensor formulation with the same random draws, on scenes that
  reach every branch of the sampler (no foreground, no background, no hard / easy background, no
  ground truth of the RoI's class, NaN RoIs, trailing padding rows).
"""
import os

import numpy as np
import pytest
import torch

from detmatch_amd import configs
from detmatch_amd.pcdet.config import ConfigDict

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
R = np.load(os.path.join(GOLD, 'pcdet_roi_head.npz'))
D = np.load(os.path.join(GOLD, 'pcdet_dense.npz'))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _head(dev):
    from detmatch_amd.pcdet.roi_heads import PVRCNNHead
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    return PVRCNNHead(input_channels=128, model_cfg=cfg.ROI_HEAD, num_class=3).to(dev)


def _golden_targets(dev):
    t = lambda k: torch.from_numpy(R[k]).to(dev)
    return {'rois': t('rois'), 'gt_of_rois': t('gt_of_rois'), 'gt_of_rois_src': t('gt_of_rois_src'),
            'reg_valid_mask': t('reg_valid_mask').long(), 'rcnn_cls_labels': t('rcnn_cls_labels'),
            'rcnn_cls': t('rcnn_cls').requires_grad_(True), 'rcnn_reg': t('rcnn_reg').requires_grad_(True)}


def test_rcnn_loss_kernel_matches_reference_values(dev):
    """roi_head_template.py:136-218: the reference's own loss values on the reference's targets."""
    h = _head(dev)
    h.forward_ret_dict = _golden_targets(dev)
    loss, tb = h.get_loss()
    close(tb['rcnn_loss_cls'], R['loss_cls'], rtol=1e-5)
    close(tb['rcnn_loss_reg'] + tb['rcnn_loss_corner'], R['loss_reg'], rtol=1e-5)
    close(tb['rcnn_loss_corner'], R['loss_corner'], rtol=1e-5)
    close(loss, float(R['loss_cls']) + float(R['loss_reg']), rtol=1e-5)
    d = _golden_targets(dev)
    d['reg_valid_mask'] = torch.zeros_like(d['reg_valid_mask'])
    h.forward_ret_dict = d
    _, tb0 = h.get_loss()
    close(tb0['rcnn_loss_reg'] + tb0['rcnn_loss_corner'], R['loss_reg_no_fg'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('weights', [(1.0, 1.0, 1.0), (0.5, 2.0, 0.25)])
def test_rcnn_loss_kernel_gradients_match_autograd(dev, weights):
    h = _head(dev)
    lw = h.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
    lw['rcnn_cls_weight'], lw['rcnn_reg_weight'], lw['rcnn_corner_weight'] = weights
    grads = []
    for fused in (True, False):
        d = _golden_targets(dev)
        # make the predictions informative: around the encoded targets, some beyond the smooth-l1 knee
        g = torch.Generator().manual_seed(3)
        d['rcnn_reg'] = (d['rcnn_reg'].detach() + 0.3 * torch.randn(d['rcnn_reg'].shape, generator=g).to(dev)
                         ).requires_grad_(True)
        h.forward_ret_dict = d
        loss, tb = h.get_loss(fused=fused)
        # upstream gradient that is not 1 (the SSL losses scale the unsupervised branch)
        (loss * 0.7).backward()
        grads.append((loss.detach(), d['rcnn_cls'].grad.clone(), d['rcnn_reg'].grad.clone(), tb))
    close(grads[0][0], grads[1][0], rtol=1e-5)
    close(grads[0][1], grads[1][1], rtol=1e-4, atol=1e-7)
    close(grads[0][2], grads[1][2], rtol=1e-4, atol=1e-7)
    for k in ('rcnn_loss_cls', 'rcnn_loss_corner', 'rcnn_loss'):
        close(grads[0][3][k], grads[1][3][k], rtol=1e-5)


def _scene(dev, seed, n_rois=512, n_gt=12, mode='mixed'):
    """RoIs scattered around ground-truth boxes so that IoUs cover [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    B = 2
    gt = torch.zeros(B, n_gt + 3, 8)
    n_real = n_gt
    gt[:, :n_real, 0] = torch.rand(B, n_real, generator=g) * 60
    gt[:, :n_real, 1] = torch.rand(B, n_real, generator=g) * 60 - 30
    gt[:, :n_real, 2] = torch.rand(B, n_real, generator=g) - 1.5
    gt[:, :n_real, 3:6] = torch.tensor([3.9, 1.6, 1.56]) * (0.8 + 0.4 * torch.rand(B, n_real, 3, generator=g))
    gt[:, :n_real, 6] = (torch.rand(B, n_real, generator=g) - 0.5) * 7
    gt[:, :n_real, 7] = torch.randint(1, 4, (B, n_real), generator=g).float()
    if mode == 'mixed':
        gt[1, 5] = 0                   # a zero row in the middle stays a candidate (only the tail is padding)
    src = torch.randint(0, n_real, (B, n_rois), generator=g)
    rois = torch.gather(gt[:, :, :7], 1, src[..., None].expand(-1, -1, 7)).clone()
    spread = {'mixed': 1.0, 'all_fg': 0.0, 'no_fg': 0.3, 'no_easy': 0.3}[mode]
    scale = torch.rand(B, n_rois, 1, generator=g) * spread
    rois[..., 0:3] += torch.randn(B, n_rois, 3, generator=g) * scale * torch.tensor([1.2, 0.6, 0.3])
    rois[..., 3:6] *= 1 + 0.15 * scale * torch.randn(B, n_rois, 3, generator=g)
    rois[..., 6] += 0.3 * scale[..., 0] * torch.randn(B, n_rois, generator=g)
    if mode == 'no_fg':
        rois[..., 3:6] *= 0.55         # IoU <= vol_roi / vol_gt: shrunken RoIs stay below REG_FG_THRESH (asserted below)
    labels = torch.gather(gt[:, :, 7], 1, src).long()
    if mode == 'mixed':
        labels[:, ::7] = (labels[:, ::7] % 3) + 1          # some RoIs carry another class
        rois[0, 3] = float('nan')                            # :109-112
        labels[1, 10:20] = 9                                 # no ground truth of this class at all
    scores = torch.rand(B, n_rois, generator=g)
    full = torch.randn(B, n_rois, 3, generator=g)
    return dict(batch_size=B, rois=rois.to(dev), roi_scores=scores.to(dev), roi_labels=labels.to(dev),
                gt_boxes=gt.to(dev), roi_scores_full=full.to(dev).requires_grad_(True))


@pytest.mark.parametrize('mode,seed', [('mixed', 0), ('mixed', 1), ('all_fg', 2), ('no_fg', 3), ('no_easy', 4)])
def test_roi_target_kernel_matches_tensor_formulation(dev, mode, seed):
    h = _head(dev)
    out = []
    for device_path in (True, False):
        bd = _scene(dev, seed, mode=mode)
        torch.manual_seed(100 + seed)
        td = h.assign_targets(bd) if device_path else h.assign_targets_tensor(bd)
        out.append(td)
    a, b = out
    ious = b['gt_iou_of_rois']
    n_fg = int((ious > 0.55).sum())
    if mode == 'all_fg':
        assert n_fg == ious.numel()
    if mode == 'no_fg':
        assert n_fg == 0
    for k in ('roi_labels', 'reg_valid_mask'):
        assert torch.equal(a[k], b[k]), k
    for k in ('rois', 'gt_of_rois_src', 'roi_scores'):
        assert torch.equal(torch.nan_to_num(a[k]), torch.nan_to_num(b[k])), k
    close(a['gt_iou_of_rois'], b['gt_iou_of_rois'], rtol=0, atol=1e-6)
    close(a['rcnn_cls_labels'], b['rcnn_cls_labels'], rtol=0, atol=2e-6)
    close(a['gt_of_rois'], b['gt_of_rois'], rtol=1e-5, atol=2e-5)
    close(a['roi_scores_full'], b['roi_scores_full'], rtol=0, atol=0)
    assert a['roi_scores_full'].requires_grad           # roi_head_template.py:98: stays in the graph
    assert a['gt_of_rois'][..., 6].abs().max() <= np.pi / 2 + 1e-6
    assert a['rois'].shape == (2, 128, 7) and a['rcnn_cls_labels'].dtype == torch.float32


def test_roi_target_sampler_statistics(dev):
    """Foreground RoIs are drawn without replacement (each at most once) and fill FG_RATIO of the
    slots; the background share of hard negatives follows HARD_BG_RATIO (:181-212)."""
    h = _head(dev)
    bd = _scene(dev, 7, mode='mixed')
    torch.manual_seed(5)
    td = h.assign_targets(bd)
    iou = td['gt_iou_of_rois']
    for b in range(2):
        fg = iou[b] >= 0.55
        assert int(fg.sum()) == 64
        fg_rois = td['rois'][b][fg]
        assert torch.unique(fg_rois, dim=0).shape[0] == 64
        hard = (iou[b] < 0.55) & (iou[b] >= 0.1)
        assert int(hard.sum()) == int(64 * 0.8)


def test_point_target_kernel_matches_tensor_formulation(dev):
    from detmatch_amd.pcdet.dense_heads import PointHeadSimple
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    for num_class in (1, 3):
        ph = PointHeadSimple(num_class=num_class, input_channels=32, model_cfg=cfg.POINT_HEAD).to(dev)
        bd = _scene(dev, 11, mode='mixed')
        gt = bd['gt_boxes']
        g = torch.Generator().manual_seed(1)
        P = 2048
        src = torch.randint(0, 12, (2, P), generator=g).to(dev)
        ctr = torch.gather(gt[:, :, :3], 1, src[..., None].expand(-1, -1, 3))
        pts = ctr + torch.randn(2, P, 3, generator=g).to(dev) * torch.tensor([1.5, 0.8, 0.6], device=dev)
        pts[:, ::5] = torch.rand(2, (P + 4) // 5, 3, generator=g).to(dev) * 60 - 20
        pts[0, 0] = 0          # the origin lies in the enlarged zero-size padding boxes
        coords = torch.cat([torch.arange(2, device=dev).repeat_interleave(P)[:, None].float(),
                            pts.reshape(-1, 3)], dim=1)
        a = ph.assign_targets(dict(point_coords=coords, gt_boxes=gt))['point_cls_labels']
        b = ph.assign_targets(dict(point_coords=coords, gt_boxes=gt), fused=False)['point_cls_labels']
        assert a.dtype == torch.int64 and torch.equal(a, b)
        assert int((a > 0).sum()) > 100 and int((a < 0).sum()) > 20 and int((a == 0).sum()) > 100


def test_point_focal_kernel_matches_reference_value_and_autograd(dev):
    from detmatch_amd.pcdet.dense_heads import PointHeadSimple
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    ph = PointHeadSimple(num_class=1, input_channels=32, model_cfg=cfg.POINT_HEAD).to(dev)
    labels = torch.from_numpy(D['ph_labels']).to(dev)
    res = []
    for fused in (True, False):
        preds = torch.from_numpy(D['ph_preds']).to(dev).requires_grad_(True)
        ph.forward_ret_dict = {'point_cls_preds': preds, 'point_cls_labels': labels}
        loss, tb = ph.get_loss(fused=fused)
        (loss * 1.3).backward()
        res.append((loss.detach(), preds.grad.clone(), tb))
    close(res[0][0], D['ph_loss'], rtol=1e-5)             # the reference's value
    close(res[0][0], res[1][0], rtol=1e-5)
    close(res[0][1], res[1][1], rtol=1e-4, atol=1e-9)
    assert float(res[0][2]['point_pos_num']) == float(D['ph_pos'])
    # three classes, labels in {-1, 0, 1, 2, 3}
    ph3 = PointHeadSimple(num_class=3, input_channels=32, model_cfg=cfg.POINT_HEAD).to(dev)
    g = torch.Generator().manual_seed(2)
    lab = torch.randint(-1, 4, (4096,), generator=g).to(dev)
    res = []
    for fused in (True, False):
        preds = (torch.randn(4096, 3, generator=torch.Generator().manual_seed(4)) * 3).to(dev).requires_grad_(True)
        ph3.forward_ret_dict = {'point_cls_preds': preds, 'point_cls_labels': lab}
        loss, _ = ph3.get_loss(fused=fused)
        loss.backward()
        res.append((loss.detach(), preds.grad.clone()))
    close(res[0][0], res[1][0], rtol=1e-5)
    close(res[0][1], res[1][1], rtol=1e-4, atol=1e-9)


S = np.load(os.path.join(GOLD, 'pcdet_roi_sampler.npz'))


@pytest.mark.parametrize('case', range(int(S['n_cases'])))
def test_roi_target_kernel_against_reference_target_layer(dev, case):
    """dm_roi_targets (max IoU with the same class, sampling, canonical transform) against the REFERENCE's
    ProposalTargetLayer.forward + RoIHeadTemplate.assign_targets (tests/golden/gen_roi_sampler_golden.py):
    every sampled row carries the reference's per-RoI IoU / assigned GT / reg_valid_mask / soft label /
    canonical GT, and the sample has the reference's foreground / hard / easy background counts (which rows
    are drawn is random in the reference too)."""
    h = _head(dev)
    k = 's%d_' % case
    bd = dict(batch_size=2, rois=torch.from_numpy(S[k + 'rois']).to(dev),
              roi_scores=torch.from_numpy(S[k + 'roi_scores']).to(dev),
              roi_labels=torch.from_numpy(S[k + 'roi_labels']).to(dev),
              gt_boxes=torch.from_numpy(S[k + 'gt_boxes']).to(dev),
              roi_scores_full=torch.from_numpy(S[k + 'roi_scores_full']).to(dev).requires_grad_(True))
    torch.manual_seed(3 + case)
    td = h.assign_targets(bd)
    rois_in = S[k + 'rois']
    got = {n: td[n].detach().cpu().numpy() for n in ('rois', 'gt_iou_of_rois', 'reg_valid_mask', 'rcnn_cls_labels',
                                                     'gt_of_rois', 'gt_of_rois_src', 'roi_labels', 'roi_scores')}
    for b in range(2):
        index = {rois_in[b, j].tobytes(): j for j in range(rois_in.shape[1])}
        src = np.array([index[r.tobytes()] for r in got['rois'][b]])            # every output row is an input row
        np.testing.assert_allclose(got['gt_iou_of_rois'][b], S[k + 'all_iou'][b, src], rtol=0, atol=2e-6)
        assert np.array_equal(got['gt_of_rois_src'][b], S[k + 'all_gt_of_rois_src'][b, src])     # assigned GT: exact
        assert np.array_equal(got['roi_labels'][b], S[k + 'roi_labels'][b, src])
        assert np.array_equal(got['roi_scores'][b], S[k + 'roi_scores'][b, src])
        # rows whose IoU sits within 2e-6 of a threshold may fall on either side; none in these scenes
        iou = S[k + 'all_iou'][b, src]
        edge = np.min(np.abs(iou[:, None] - np.array([0.1, 0.25, 0.55, 0.75])[None]), 1) < 1e-5
        assert not edge.any()
        assert np.array_equal(got['reg_valid_mask'][b], S[k + 'all_reg_valid'][b, src])
        np.testing.assert_allclose(got['rcnn_cls_labels'][b], S[k + 'all_cls_labels'][b, src], rtol=0, atol=5e-6)
        np.testing.assert_allclose(got['gt_of_rois'][b], S[k + 'all_gt_of_rois'][b, src], rtol=1e-5, atol=3e-5)
        # the sample's structure == the reference's sample
        g_iou = got['gt_iou_of_rois'][b]
        assert int((g_iou >= 0.55).sum()) == int(S[k + 'smp_n_fg'][b])
        assert int(((g_iou < 0.55) & (g_iou >= 0.1)).sum()) == int(S[k + 'smp_n_hard'][b])
        assert int((g_iou < 0.1).sum()) == int(S[k + 'smp_n_easy'][b])
        fg = src[g_iou >= 0.55]
        assert len(np.unique(fg)) == len(fg)                                   # foreground: without replacement


def test_roi_decode_kernel_equals_tensor_formulation(dev):
    """generate_predicted_boxes (roi_head_template.py:233-263) as one launch == decode_torch + rotate_points_along_z + shift,
    values and the gradient w.r.t. the refinements (fp32 rounding: the tensor chain rotates through a batched matmul)."""
    from detmatch_amd import fused
    from detmatch_amd.pcdet import utils as U
    from detmatch_amd.pcdet.roi_heads import _RoiDecode
    g = torch.Generator(device='cpu').manual_seed(0)
    n = 300
    rois = torch.cat([torch.randn(n, 3, generator=g) * 20, torch.rand(n, 3, generator=g) * 4 + 0.5,
                      (torch.rand(n, 1, generator=g) - 0.5) * 7], 1).to(dev)
    reg0 = (torch.randn(n, 7, generator=g) * 0.3).to(dev)
    go = torch.randn(n, 7, generator=g).to(dev)
    coder = U.ResidualCoder()

    def tensor_path(reg):
        local = rois.clone()
        local[:, 0:3] = 0
        d = coder.decode_torch(reg.view(1, -1, 7), local.view(1, -1, 7)).view(-1, 7)
        d = U.rotate_points_along_z(d.unsqueeze(1), rois[:, 6]).squeeze(1)
        return torch.cat([d[:, 0:3] + rois[:, 0:3], d[:, 3:]], dim=-1)
    ra = reg0.clone().requires_grad_(True)
    want = tensor_path(ra)
    want.backward(go)
    rb = reg0.clone().requires_grad_(True)
    got = _RoiDecode.apply(rb, rois)
    got.backward(go)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float((rb.grad - ra.grad).abs().max()) <= 2e-5 * float(ra.grad.abs().max())


def test_roi_grid_points_kernel_equals_tensor_formulation(dev):
    """dm_roi_grid_points == PVRCNNHead.get_global_grid_points_of_roi (pvrcnn_head.py:127-149) to fp32 rounding (the tensor
    chain rotates through a batched matmul), same point order."""
    from detmatch_amd import _lib
    from detmatch_amd.pcdet.roi_heads import PVRCNNHead
    g = torch.Generator(device='cpu').manual_seed(3)
    n, gs = 77, 6
    rois = torch.cat([torch.randn(n, 3, generator=g) * 30, torch.rand(n, 3, generator=g) * 5 + 0.3,
                      (torch.rand(n, 1, generator=g) - 0.5) * 7], 1).to(dev)
    want, _ = PVRCNNHead.get_global_grid_points_of_roi(PVRCNNHead, rois.view(1, n, 7), gs)
    got = torch.empty((n * gs ** 3, 3), device=dev)
    _lib.check(_lib.lib().dm_roi_grid_points(rois.data_ptr(), n, 7, gs, got.data_ptr(), _lib.stream()), 'grid')
    assert float((got.view(n, -1, 3) - want).abs().max()) <= 1e-5 * float(want.abs().max())
