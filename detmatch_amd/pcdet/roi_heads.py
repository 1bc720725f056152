"""PVRCNNHead: proposal layer (rotated NMS), proposal-target sampling, RoI-grid pooling,
box refinement, losses.

Reference: pcdet/models/roi_heads/{roi_head_template,pvrcnn_head}.py,
roi_heads/target_assigner/proposal_target_layer.py, model_utils/model_nms_utils.py.

Restated without the reference's host round-trips: NMS keeps its survivor list on the
device (fixed-size, masked), same-class max-IoU is a dense masked max instead of a
`.item()` class loop (:237), RoI sub-sampling draws its random numbers on the device
(same distribution as the numpy/torch-CPU draws at :153,:164,:193-197 — a different
stream, as any two seeds are), `tb_dict` entries are detached tensors.
`roi_scores_full` stays attached to the graph (roi_head_template.py:98).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, iou3d_nms
from ..devconst import const
from .. import pointnet2_stack as pn2
from . import utils as U
from ..bn_relu import fc_rows
from ..fused import on as fused_on
from .dense_heads import valid_gt_mask
from .pfe import batch_row_counts


def class_agnostic_nms_fixed_batch(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.py:6-26 (class_agnostic_nms) for every sample of a batch with a fixed-size result
    and no device->host copy: ONE top-k, sort and gather over the (B, N) score matrix (the per-sample
    loops of roi_head_template.py:60-90 and detector3d_template.py:176-309 repeat the same ~50 launches
    per sample); the NMS itself stays one call per sample.  -> selected (B, POST) int64 padded with 0,
    valid (B, POST) bool."""
    post = int(nms_config.NMS_POST_MAXSIZE)
    bsz, n = box_scores.shape
    dev = box_scores.device
    if n == 0 or bsz == 0:
        return (torch.zeros((bsz, post), dtype=torch.int64, device=dev),
                torch.zeros((bsz, post), dtype=torch.bool, device=dev))
    scores = box_scores
    if score_thresh is not None:  # below-threshold boxes sort last and are masked out
        scores = torch.where(box_scores >= score_thresh, box_scores, box_scores.new_full((), -float('inf')))
    k = min(int(nms_config.NMS_PRE_MAXSIZE), n)
    # torch.topk(sorted=True) hands the k scores over in descending order: the (stable) `scores.sort(descending=True)` that
    # nms_gpu / nms_normal_gpu start with (iou3d_nms_utils.py:66,90) is the identity on it — no second sort (it was a
    # 95 us launch for 9 000 candidates on the main lane, three times per iteration)
    top_scores, indices = torch.topk(scores, k=k, dim=1, largest=True, sorted=True)      # original index, NMS order
    boxes = torch.gather(box_preds[:, :, 0:7], 1, indices[:, :, None].expand(-1, -1, 7)).contiguous().float()
    L = _lib.lib()
    keep = torch.zeros((bsz, max(k, post)), dtype=torch.int64, device=dev)
    num = torch.zeros((bsz,), dtype=torch.int32, device=dev)
    # all samples in one launch chain (the greedy pass is one wave per sample)
    ws = _lib.workspace(L.dm_nms_workspace_bytes(k) * bsz, dev, 'nms')
    _lib.check(L.dm_nms_batch(_lib.ptr(boxes), bsz, k, float(nms_config.NMS_THRESH), post,
                              0 if nms_config.NMS_TYPE == 'nms_gpu' else 1, _lib.ptr(keep), keep.stride(0),
                              _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_nms_batch')
    valid = torch.arange(post, device=dev)[None, :] < num.long()[:, None]
    sel = torch.gather(indices, 1, keep[:, :post].clamp(0, k - 1))
    if score_thresh is not None:
        valid = valid & (torch.gather(box_scores, 1, sel) >= score_thresh)
    return torch.where(valid, sel, torch.zeros_like(sel)), valid


class ProposalTargetLayer(nn.Module):
    """proposal_target_layer.py:8-259"""

    def __init__(self, roi_sampler_cfg):
        super().__init__()
        self.roi_sampler_cfg = roi_sampler_cfg

    def forward(self, batch_dict):
        cfg = self.roi_sampler_cfg
        (batch_rois, batch_gt_of_rois, batch_roi_ious, batch_roi_scores, batch_roi_labels,
         batch_roi_scores_full) = self.sample_rois_for_rcnn(batch_dict)
        reg_valid_mask = (batch_roi_ious > cfg.REG_FG_THRESH).long()
        assert cfg.CLS_SCORE_TYPE == 'roi_iou'
        iou_bg, iou_fg = cfg.CLS_BG_THRESH, cfg.CLS_FG_THRESH
        fg_mask = batch_roi_ious > iou_fg
        bg_mask = batch_roi_ious < iou_bg
        interval = (~fg_mask) & (~bg_mask)
        cls_labels = torch.where(interval, (batch_roi_ious - iou_bg) / (iou_fg - iou_bg),
                                 fg_mask.float())
        return {'rois': batch_rois.detach(), 'gt_of_rois': batch_gt_of_rois.detach(),
                'gt_iou_of_rois': batch_roi_ious.detach(), 'roi_scores': batch_roi_scores.detach(),
                'roi_labels': batch_roi_labels.detach(), 'roi_scores_full': batch_roi_scores_full,
                'reg_valid_mask': reg_valid_mask.detach(), 'rcnn_cls_labels': cls_labels.detach()}

    def forward_device(self, batch_dict):
        """forward() and the canonical transform of RoIHeadTemplate.assign_targets as two launches
        (dm_roi_targets, csrc/roi_targets.hip); the same random draws as the tensor formulation."""
        from .. import _lib
        cfg = self.roi_sampler_cfg
        assert cfg.CLS_SCORE_TYPE == 'roi_iou'
        rois = batch_dict['rois'].detach().float().contiguous()
        scores = batch_dict['roi_scores'].detach().float().contiguous()
        labels = batch_dict['roi_labels'].detach().long().contiguous()
        gt = batch_dict['gt_boxes'].detach().float().contiguous()
        scores_full = batch_dict['roi_scores_full']
        _lib.require_device(rois, scores, labels, gt)
        B, R = int(rois.shape[0]), int(rois.shape[1])
        G, gtc = int(gt.shape[1]), int(gt.shape[2])
        S = int(cfg.ROI_PER_IMAGE)
        u_perm, u_pick = self.draw(B, R, rois.device)
        dev = rois.device
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        i64 = lambda *shape: torch.empty(shape, dtype=torch.int64, device=dev)
        o_rois, o_src, o_ct = f32(B, S, 7), f32(B, S, gtc), f32(B, S, gtc)
        o_iou, o_score, o_cls, o_ok = f32(B, S), f32(B, S), f32(B, S), f32(B)
        o_label, o_valid, o_sampled = i64(B, S), i64(B, S), i64(B, S)
        L = _lib.lib()
        ws = _lib.workspace(L.dm_roi_targets_workspace_bytes(B, R), dev, 'roi_targets')
        _lib.check(L.dm_roi_targets(
            _lib.ptr(rois), _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(gt), B, R, G, gtc,
            _lib.ptr(u_perm), _lib.ptr(u_pick), S, int(np.round(cfg.FG_RATIO * S)),
            float(cfg.REG_FG_THRESH), float(cfg.CLS_FG_THRESH), float(cfg.CLS_BG_THRESH),
            float(cfg.CLS_BG_THRESH_LO), float(cfg.HARD_BG_RATIO), _lib.ptr(o_rois), _lib.ptr(o_src),
            _lib.ptr(o_ct), _lib.ptr(o_iou), _lib.ptr(o_score), _lib.ptr(o_label), _lib.ptr(o_valid),
            _lib.ptr(o_cls), _lib.ptr(o_sampled), _lib.ptr(o_ok), _lib.ptr(ws), ws.numel(),
            _lib.stream()), 'dm_roi_targets')
        # the one tensor of the layer that stays in the autograd graph (roi_head_template.py:98)
        full = torch.gather(scores_full, 1, o_sampled[..., None].expand(-1, -1, scores_full.shape[-1])
                            ) * o_ok.view(-1, 1, 1)
        return {'rois': o_rois, 'gt_of_rois': o_ct, 'gt_of_rois_src': o_src, 'gt_iou_of_rois': o_iou,
                'roi_scores': o_score, 'roi_labels': o_label, 'roi_scores_full': full,
                'reg_valid_mask': o_valid, 'rcnn_cls_labels': o_cls}

    def sample_rois_for_rcnn(self, batch_dict):
        """:69-134, batched over the samples."""
        rois = batch_dict['rois']
        roi_scores = batch_dict['roi_scores']
        roi_labels = batch_dict['roi_labels']
        gt_boxes = batch_dict['gt_boxes']
        roi_scores_full = batch_dict['roi_scores_full']
        rois = torch.where(torch.isnan(rois), torch.zeros_like(rois), rois)     # :109-112
        gt_valid = valid_gt_mask(gt_boxes)                                       # :101-104
        max_overlaps, gt_assignment = self.get_max_iou_with_same_class(
            rois, roi_labels, gt_boxes[:, :, 0:7], gt_boxes[:, :, -1].long(), gt_valid)
        u_perm, u_pick = self.draw(rois.shape[0], rois.shape[1], rois.device)
        sampled, ok = self.subsample_rois(max_overlaps, u_perm, u_pick)          # (B, S), (B,)
        okf = ok.view(-1, 1)
        g = lambda t: torch.gather(t, 1, sampled)
        batch_rois = torch.gather(rois, 1, sampled[..., None].expand(-1, -1, rois.shape[-1]))
        batch_rois = batch_rois * okf[..., None].to(rois.dtype)
        batch_roi_labels = g(roi_labels) * okf.long()
        batch_roi_ious = g(max_overlaps) * okf.to(rois.dtype)
        batch_roi_scores = g(roi_scores) * okf.to(rois.dtype)
        batch_roi_scores_full = torch.gather(
            roi_scores_full, 1, sampled[..., None].expand(-1, -1, roi_scores_full.shape[-1])
        ) * okf[..., None].to(rois.dtype)
        ga = g(gt_assignment)
        batch_gt_of_rois = torch.gather(gt_boxes, 1, ga[..., None].expand(-1, -1, gt_boxes.shape[-1]))
        batch_gt_of_rois = batch_gt_of_rois * okf[..., None].to(rois.dtype)
        return (batch_rois, batch_gt_of_rois, batch_roi_ious, batch_roi_scores, batch_roi_labels,
                batch_roi_scores_full)

    @staticmethod
    def get_max_iou_with_same_class(rois, roi_labels, gt_boxes, gt_labels, gt_valid):
        """:217-259 as a dense masked max: rois (B,R,7), gt (B,G,7)."""
        B = rois.shape[0]
        mo, ga = [], []
        for b in range(B):
            iou3d = iou3d_nms.boxes_iou3d_gpu(rois[b], gt_boxes[b])              # (R, G)
            same = (roi_labels[b][:, None] == gt_labels[b][None, :]) & gt_valid[b][None, :]
            iou_m = torch.where(same, iou3d, iou3d.new_full((), -1.0))
            m, a = iou_m.max(dim=1)
            has = m >= 0
            mo.append(torch.where(has, m, torch.zeros_like(m)))
            ga.append(torch.where(has, a, torch.zeros_like(a)))
        return torch.stack(mo), torch.stack(ga)

    def draw(self, batch, n_rois, device):
        """The sampler's random numbers (np.random.permutation / randint / rand in the reference,
        :153,:185-199): one uniform key per RoI (orders the foreground set) and one uniform per
        output slot (picks with replacement)."""
        return (torch.rand((batch, n_rois), device=device),
                torch.rand((batch, int(self.roi_sampler_cfg.ROI_PER_IMAGE)), device=device))

    def subsample_rois(self, max_overlaps, u_perm, u):
        """:136-215 for all samples at once.  Returns indices (B, ROI_PER_IMAGE) and a
        per-sample flag that is False only in the reference's 'no fg and no bg' error case."""
        cfg = self.roi_sampler_cfg
        B, R = max_overlaps.shape
        S = int(cfg.ROI_PER_IMAGE)
        dev = max_overlaps.device
        fg_per_image = int(np.round(cfg.FG_RATIO * S))
        fg_thresh = min(cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH)
        fg = max_overlaps >= fg_thresh
        easy = max_overlaps < cfg.CLS_BG_THRESH_LO
        hard = (max_overlaps < cfg.REG_FG_THRESH) & (max_overlaps >= cfg.CLS_BG_THRESH_LO)
        n_fg, n_easy, n_hard = fg.sum(1), easy.sum(1), hard.sum(1)
        n_bg = n_easy + n_hard
        # members of each set in ascending index order come first
        ar = torch.arange(R, device=dev)[None, :].expand(B, -1)
        rank_of = lambda m: torch.sort(torch.where(m, ar, ar + R), dim=1)[1]
        fg_sorted, easy_sorted, hard_sorted = rank_of(fg), rank_of(easy), rank_of(hard)
        # random permutation of the fg members (np.random.permutation, :153)
        key = torch.where(fg, u_perm, torch.full((B, R), 2.0, device=dev))
        fg_perm = torch.sort(key, dim=1, stable=True)[1]
        slot = torch.arange(S, device=dev)[None, :].expand(B, -1)
        pick = lambda srt, n: torch.gather(
            srt, 1, torch.minimum((u * n[:, None]).long(), (n[:, None] - 1).clamp(min=0)))
        fg_this = torch.minimum(n_fg, torch.full_like(n_fg, fg_per_image))
        fg_this = torch.where(n_bg > 0, fg_this, torch.full_like(n_fg, S))       # :167-172
        fg_this = torch.where(n_fg > 0, fg_this, torch.zeros_like(n_fg))         # :174-179
        bg_this = S - fg_this
        hard_num = torch.minimum((bg_this.float() * cfg.HARD_BG_RATIO).long(), n_hard)
        hard_num = torch.where(n_easy > 0, hard_num, bg_this)                    # :203-207
        hard_num = torch.where(n_hard > 0, hard_num, torch.zeros_like(hard_num))  # :208-212
        fg_wo_replace = torch.gather(fg_perm, 1, slot.clamp(max=R - 1))
        fg_choice = torch.where((n_bg > 0)[:, None], fg_wo_replace, pick(fg_sorted, n_fg))
        j = slot - fg_this[:, None]
        bg_choice = torch.where(j < hard_num[:, None], pick(hard_sorted, n_hard),
                                pick(easy_sorted, n_easy))
        sampled = torch.where(slot < fg_this[:, None], fg_choice, bg_choice)
        ok = (n_fg + n_bg) > 0
        return sampled, ok


class _FusedRcnnLoss(torch.autograd.Function):
    """dm_rcnn_loss_forward / _backward (csrc/roi_targets.hip): classification, smooth-l1 and corner
    losses of the RoI head in one launch -> tensor [cls, reg, corner]; the backward is one launch."""

    @staticmethod
    def forward(ctx, rcnn_cls, rcnn_reg, rois, gt_ct, gt_src, reg_valid, cls_labels, meta):
        L = _lib.lib()
        _lib.require_device(rcnn_cls, rcnn_reg, rois, gt_ct, gt_src, reg_valid, cls_labels)
        w3, cw7, beta, corner = meta
        n = int(rcnn_reg.shape[0])
        dev = rcnn_reg.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        g_cls = torch.empty(n, dtype=torch.float32, device=dev)
        g_sl1 = torch.empty((n, 7), dtype=torch.float32, device=dev)
        g_corner = torch.empty((n, 7), dtype=torch.float32, device=dev)
        _lib.check(L.dm_rcnn_loss_forward(
            _lib.ptr(rcnn_cls), _lib.ptr(rcnn_reg), _lib.ptr(rois), _lib.ptr(gt_ct), _lib.ptr(gt_src),
            _lib.ptr(reg_valid), _lib.ptr(cls_labels), n, int(gt_ct.shape[-1]), _lib.floats(w3),
            _lib.floats(cw7), float(beta), int(corner), _lib.ptr(out), _lib.ptr(g_cls), _lib.ptr(g_sl1),
            _lib.ptr(g_corner), _lib.stream()), 'dm_rcnn_loss_forward')
        ctx.save_for_backward(g_cls, g_sl1, g_corner)
        ctx.cls_shape = rcnn_cls.shape
        return out

    @staticmethod
    def backward(ctx, grad):
        g_cls, g_sl1, g_corner = ctx.saved_tensors
        n = int(g_cls.shape[0])
        grad = grad.contiguous().float()
        d_cls, d_reg = torch.empty_like(g_cls), torch.empty_like(g_sl1)
        _lib.check(_lib.lib().dm_rcnn_loss_backward(
            _lib.ptr(grad), _lib.ptr(g_cls), _lib.ptr(g_sl1), _lib.ptr(g_corner), n, _lib.ptr(d_cls),
            _lib.ptr(d_reg), _lib.stream()), 'dm_rcnn_loss_backward')
        return d_cls.view(ctx.cls_shape), d_reg, None, None, None, None, None, None


class _RoiDecode(torch.autograd.Function):
    """dm_roi_decode_forward / _backward (csrc/box_decode.hip): generate_predicted_boxes' decode + rotate + translate in
    one launch, gradient w.r.t. the refinements in one launch."""

    @staticmethod
    def forward(ctx, reg, rois):
        reg = reg.contiguous()
        rois = rois.detach().contiguous()
        _lib.require_device(reg, rois)
        n = int(reg.shape[0])
        out = torch.empty((n, 7), dtype=torch.float32, device=reg.device)
        _lib.check(_lib.lib().dm_roi_decode_forward(reg.data_ptr(), rois.data_ptr(), n, out.data_ptr(), _lib.raw_stream()),
                   'dm_roi_decode_forward')
        ctx.save_for_backward(reg, rois)
        return out

    @staticmethod
    def backward(ctx, grad):
        reg, rois = ctx.saved_tensors
        grad = grad.contiguous()
        g = torch.empty_like(reg)
        _lib.check(_lib.lib().dm_roi_decode_backward(grad.data_ptr(), reg.data_ptr(), rois.data_ptr(), int(reg.shape[0]),
                                                     g.data_ptr(), _lib.raw_stream()), 'dm_roi_decode_backward')
        return g, None


class PVRCNNHead(nn.Module):
    """roi_head_template.py:11-263 + pvrcnn_head.py:8-201."""

    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        assert model_cfg.TARGET_CONFIG.BOX_CODER == 'ResidualCoder'
        self.box_coder = U.ResidualCoder(**model_cfg.TARGET_CONFIG.get('BOX_CODER_CONFIG', {}))
        self.proposal_target_layer = ProposalTargetLayer(roi_sampler_cfg=model_cfg.TARGET_CONFIG)
        self.reg_loss_func = U.WeightedSmoothL1Loss(
            code_weights=model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['code_weights'])
        self.forward_ret_dict = None
        mlps = [[input_channels] + list(m) for m in model_cfg.ROI_GRID_POOL.MLPS]
        self.roi_grid_pool_layer = pn2.StackSAModuleMSG(
            radii=model_cfg.ROI_GRID_POOL.POOL_RADIUS, nsamples=model_cfg.ROI_GRID_POOL.NSAMPLE,
            mlps=mlps, use_xyz=True, pool_method=model_cfg.ROI_GRID_POOL.POOL_METHOD)
        gs = model_cfg.ROI_GRID_POOL.GRID_SIZE
        pre_channel = gs * gs * gs * sum([x[-1] for x in mlps])
        shared = []
        for k in range(len(model_cfg.SHARED_FC)):
            shared.extend([nn.Conv1d(pre_channel, model_cfg.SHARED_FC[k], kernel_size=1, bias=False),
                           nn.BatchNorm1d(model_cfg.SHARED_FC[k]), nn.ReLU()])
            pre_channel = model_cfg.SHARED_FC[k]
            if k != len(model_cfg.SHARED_FC) - 1 and model_cfg.DP_RATIO > 0:
                shared.append(nn.Dropout(model_cfg.DP_RATIO))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = self.make_fc_layers(pre_channel, self.num_class, model_cfg.CLS_FC)
        self.reg_layers = self.make_fc_layers(pre_channel, self.box_coder.code_size * self.num_class,
                                              model_cfg.REG_FC)
        for m in self.modules():  # init_weights('xavier'), pvrcnn_head.py:54-71
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    def make_fc_layers(self, input_channels, output_channels, fc_list):
        layers = []
        pre = input_channels
        for k in range(len(fc_list)):
            layers.extend([nn.Conv1d(pre, fc_list[k], kernel_size=1, bias=False),
                           nn.BatchNorm1d(fc_list[k]), nn.ReLU()])
            pre = fc_list[k]
            if self.model_cfg.DP_RATIO >= 0 and k == 0:
                layers.append(nn.Dropout(self.model_cfg.DP_RATIO))
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    # ---- proposals ------------------------------------------------------------
    def proposal_layer(self, batch_dict, nms_config):
        """roi_head_template.py:46-102"""
        batch_box_preds = batch_dict['batch_box_preds']
        batch_cls_preds = batch_dict['batch_cls_preds']
        assert batch_cls_preds.dim() == 3 and not nms_config.MULTI_CLASSES_NMS
        scores, labels = torch.max(batch_cls_preds, dim=2)
        sel, valid = class_agnostic_nms_fixed_batch(scores.detach(), batch_box_preds.detach(), nms_config)
        vf = valid.to(batch_box_preds.dtype)
        pick = lambda t: torch.gather(t, 1, sel[:, :, None].expand(-1, -1, t.shape[2]))   # noqa: E731
        batch_dict['rois'] = (pick(batch_box_preds) * vf[:, :, None]).detach()
        batch_dict['roi_scores'] = (torch.gather(scores, 1, sel) * vf).detach()
        batch_dict['roi_scores_full'] = pick(batch_cls_preds) * vf[:, :, None]   # not detached (:98)
        batch_dict['roi_labels'] = (torch.gather(labels, 1, sel) * valid.long() + 1).detach()
        batch_dict['has_class_labels'] = batch_cls_preds.shape[-1] > 1
        batch_dict.pop('batch_index', None)
        return batch_dict

    def assign_targets(self, batch_dict):
        """roi_head_template.py:104-134: sample + canonical transform of the GT."""
        if fused_on() and batch_dict['rois'].is_cuda and batch_dict['gt_boxes'].shape[1] > 0 \
                and 8 <= batch_dict['gt_boxes'].shape[2] <= 16 and batch_dict['rois'].shape[1] <= 3072:
            return self.proposal_target_layer.forward_device(batch_dict)
        return self.assign_targets_tensor(batch_dict)

    def assign_targets_tensor(self, batch_dict):
        """The same as a chain of tensor operations (host-logic tests on CPU; comparator of the kernel)."""
        batch_size = batch_dict['batch_size']
        targets_dict = self.proposal_target_layer(batch_dict)
        rois = targets_dict['rois']
        gt_of_rois = targets_dict['gt_of_rois']
        targets_dict['gt_of_rois_src'] = gt_of_rois.clone().detach()
        roi_center = rois[:, :, 0:3]
        roi_ry = rois[:, :, 6] % (2 * np.pi)
        gt_of_rois = torch.cat([gt_of_rois[:, :, 0:3] - roi_center, gt_of_rois[:, :, 3:6],
                                (gt_of_rois[:, :, 6] - roi_ry).unsqueeze(-1), gt_of_rois[:, :, 7:]],
                               dim=-1)
        gt_of_rois = U.rotate_points_along_z(gt_of_rois.view(-1, 1, gt_of_rois.shape[-1]),
                                             -roi_ry.view(-1)).view(batch_size, -1,
                                                                    gt_of_rois.shape[-1])
        heading = gt_of_rois[:, :, 6] % (2 * np.pi)
        opposite = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading = torch.where(opposite, (heading + np.pi) % (2 * np.pi), heading)
        heading = torch.where(heading > np.pi, heading - np.pi * 2, heading)
        heading = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        targets_dict['gt_of_rois'] = torch.cat([gt_of_rois[:, :, :6], heading.unsqueeze(-1),
                                                gt_of_rois[:, :, 7:]], dim=-1)
        return targets_dict

    # ---- RoI grid pooling -------------------------------------------------------
    @staticmethod
    def get_dense_grid_points(rois, batch_size_rcnn, grid_size):
        """pvrcnn_head.py:140-149"""
        g = np.arange(grid_size)
        idx = np.stack(np.meshgrid(g, g, g, indexing='ij'), axis=-1).reshape(-1, 3)   # == ones.nonzero()
        dense_idx = const(idx.astype(np.float32), rois.device).repeat(batch_size_rcnn, 1, 1)
        local_roi_size = rois.view(batch_size_rcnn, -1)[:, 3:6]
        return (dense_idx + 0.5) / grid_size * local_roi_size.unsqueeze(dim=1) \
            - (local_roi_size.unsqueeze(dim=1) / 2)

    def get_global_grid_points_of_roi(self, rois, grid_size):
        rois = rois.view(-1, rois.shape[-1])
        local = self.get_dense_grid_points(rois, rois.shape[0], grid_size)
        glob = U.rotate_points_along_z(local.clone(), rois[:, 6]).squeeze(dim=1)
        return glob + rois[:, 0:3].clone().unsqueeze(dim=1), local

    def roi_grid_pool(self, batch_dict):
        """pvrcnn_head.py:73-125"""
        batch_size = batch_dict['batch_size']
        rois = batch_dict['rois']
        point_coords = batch_dict['point_coords']
        point_features = batch_dict['point_features'] * batch_dict['point_cls_scores'].view(-1, 1)
        gs = self.model_cfg.ROI_GRID_POOL.GRID_SIZE
        if fused_on() and rois.is_cuda and rois.dtype == torch.float32:
            flat = rois.detach().reshape(-1, rois.shape[-1]).contiguous()
            global_pts = torch.empty((flat.shape[0] * gs ** 3, 3), dtype=torch.float32, device=rois.device)
            _lib.check(_lib.lib().dm_roi_grid_points(flat.data_ptr(), int(flat.shape[0]), int(flat.shape[1]), int(gs),
                                                     global_pts.data_ptr(), _lib.raw_stream()), 'dm_roi_grid_points')
        else:
            global_pts, _ = self.get_global_grid_points_of_roi(rois, grid_size=gs)
        global_pts = global_pts.view(batch_size, -1, 3)
        xyz = point_coords[:, 1:4]
        xyz_batch_cnt = batch_dict.get('point_batch_cnt')      # set by the key-point encoder (rows per sample)
        if xyz_batch_cnt is None or xyz_batch_cnt.numel() != batch_size:
            xyz_batch_cnt = batch_row_counts(point_coords[:, 0], batch_size)
        new_xyz = global_pts.view(-1, 3)
        new_xyz_batch_cnt = torch.full((batch_size,), global_pts.shape[1], dtype=torch.int32,
                                       device=xyz.device)
        _, pooled = self.roi_grid_pool_layer(xyz=xyz.contiguous(), xyz_batch_cnt=xyz_batch_cnt,
                                             new_xyz=new_xyz.contiguous(),
                                             new_xyz_batch_cnt=new_xyz_batch_cnt,
                                             features=point_features.contiguous())
        return pooled.view(-1, gs ** 3, pooled.shape[-1])

    def forward(self, batch_dict):
        """pvrcnn_head.py:151-201"""
        self.proposal_layer(batch_dict,
                            nms_config=self.model_cfg.NMS_CONFIG['TRAIN' if self.training else 'TEST'])
        targets_dict = None
        if self.training:
            targets_dict = self.assign_targets(batch_dict)
            batch_dict['rois'] = targets_dict['rois']
            batch_dict['roi_labels'] = targets_dict['roi_labels']
            batch_dict['roi_scores'] = targets_dict['roi_scores']
            batch_dict['roi_scores_full'] = targets_dict['roi_scores_full']
        pooled = self.roi_grid_pool(batch_dict)                           # (B*N, 216, C)
        n_rcnn = pooled.shape[0]
        pooled = pooled.permute(0, 2, 1).contiguous()                    # (B*N, C, 216)
        shared = fc_rows(self.shared_fc_layer, pooled.view(n_rcnn, -1))
        rcnn_cls = fc_rows(self.cls_layers, shared)                # (B*N, 1 or num_class)
        rcnn_reg = fc_rows(self.reg_layers, shared)                # (B*N, code_size * ...)
        batch_cls_preds, batch_box_preds = self.generate_predicted_boxes(
            batch_dict['batch_size'], batch_dict['rois'], rcnn_cls, rcnn_reg)
        batch_dict['batch_cls_preds'] = batch_cls_preds
        batch_dict['batch_box_preds'] = batch_box_preds
        batch_dict['cls_preds_normalized'] = False
        if self.training:
            targets_dict['rcnn_cls'] = rcnn_cls
            targets_dict['rcnn_reg'] = rcnn_reg
            self.forward_ret_dict = targets_dict
        return batch_dict

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        """roi_head_template.py:233-263"""
        code_size = self.box_coder.code_size
        batch_cls_preds = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        if fused_on() and box_preds.is_cuda and code_size == 7 and rois.shape[-1] == 7 and box_preds.dtype == torch.float32 \
                and rois.dtype == torch.float32 and box_preds.numel() == rois.numel():
            decoded = _RoiDecode.apply(box_preds.view(-1, 7), rois.view(-1, 7))
            return batch_cls_preds, decoded.view(batch_size, -1, code_size)
        batch_box_preds = box_preds.view(batch_size, -1, code_size)
        roi_ry = rois[:, :, 6].view(-1)
        roi_xyz = rois[:, :, 0:3].view(-1, 3)
        local_rois = rois.clone().detach()
        local_rois[:, :, 0:3] = 0
        decoded = self.box_coder.decode_torch(batch_box_preds, local_rois).view(-1, code_size)
        decoded = U.rotate_points_along_z(decoded.unsqueeze(dim=1), roi_ry).squeeze(dim=1)
        decoded = torch.cat([decoded[:, 0:3] + roi_xyz, decoded[:, 3:]], dim=-1)
        return batch_cls_preds, decoded.view(batch_size, -1, code_size)

    # ---- losses ---------------------------------------------------------------
    def get_box_reg_layer_loss(self, d):
        """roi_head_template.py:136-198 (smooth-l1 + corner regularisation), with the fg
        subset expressed as a mask instead of a data-dependent gather."""
        loss_cfgs = self.model_cfg.LOSS_CONFIG
        assert loss_cfgs.REG_LOSS == 'smooth-l1'
        code_size = self.box_coder.code_size
        reg_valid_mask = d['reg_valid_mask'].view(-1)
        gt_boxes3d_ct = d['gt_of_rois'][..., 0:code_size]
        gt_of_rois_src = d['gt_of_rois_src'][..., 0:code_size].view(-1, code_size)
        rcnn_reg = d['rcnn_reg']
        roi_boxes3d = d['rois']
        n = gt_boxes3d_ct.view(-1, code_size).shape[0]
        fg_mask = reg_valid_mask > 0
        fgf = fg_mask.float()
        fg_sum = fgf.sum()
        rois_anchor = roi_boxes3d.clone().detach().view(-1, code_size)
        rois_anchor[:, 0:3] = 0
        rois_anchor[:, 6] = 0
        reg_targets = self.box_coder.encode_torch(gt_boxes3d_ct.view(n, code_size), rois_anchor)
        loss_reg = self.reg_loss_func(rcnn_reg.view(n, -1).unsqueeze(dim=0),
                                      reg_targets.unsqueeze(dim=0))
        # rows that are not fg may hold inf/nan targets (zero-size boxes): mask, don't multiply
        loss_reg = torch.where(fg_mask[:, None], loss_reg.view(n, -1),
                               torch.zeros_like(loss_reg.view(n, -1))).sum() \
            / torch.clamp(fg_sum, min=1.0)
        loss_reg = loss_reg * loss_cfgs.LOSS_WEIGHTS['rcnn_reg_weight']
        tb_dict = {'rcnn_loss_reg': loss_reg.detach()}
        if loss_cfgs.CORNER_LOSS_REGULARIZATION:
            rois_flat = roi_boxes3d.view(-1, code_size)
            batch_anchors = rois_flat.clone().detach().view(1, -1, code_size)
            roi_ry = rois_flat[:, 6]
            roi_xyz = rois_flat[:, 0:3]
            batch_anchors[:, :, 0:3] = 0
            rcnn_boxes3d = self.box_coder.decode_torch(rcnn_reg.view(1, -1, code_size),
                                                       batch_anchors).view(-1, code_size)
            rcnn_boxes3d = U.rotate_points_along_z(rcnn_boxes3d.unsqueeze(dim=1), roi_ry).squeeze(dim=1)
            rcnn_boxes3d = torch.cat([rcnn_boxes3d[:, 0:3] + roi_xyz, rcnn_boxes3d[:, 3:]], dim=-1)
            corner = U.get_corner_loss_lidar(rcnn_boxes3d[:, 0:7], gt_of_rois_src[:, 0:7])
            corner = torch.where(fg_mask, corner, torch.zeros_like(corner)).sum() \
                / torch.clamp(fg_sum, min=1.0)     # == mean over the fg rows; 0 when there are none
            corner = corner * loss_cfgs.LOSS_WEIGHTS['rcnn_corner_weight']
            loss_reg = loss_reg + corner
            tb_dict['rcnn_loss_corner'] = corner.detach()
        return loss_reg, tb_dict

    def get_box_cls_layer_loss(self, d):
        """roi_head_template.py:200-218"""
        loss_cfgs = self.model_cfg.LOSS_CONFIG
        assert loss_cfgs.CLS_LOSS == 'BinaryCrossEntropy'
        rcnn_cls_flat = d['rcnn_cls'].view(-1)
        labels = d['rcnn_cls_labels'].view(-1)
        batch_loss_cls = F.binary_cross_entropy(torch.sigmoid(rcnn_cls_flat), labels.float(),
                                                reduction='none')
        valid = (labels >= 0).float()
        loss = (batch_loss_cls * valid).sum() / torch.clamp(valid.sum(), min=1.0)
        loss = loss * loss_cfgs.LOSS_WEIGHTS['rcnn_cls_weight']
        return loss, {'rcnn_loss_cls': loss.detach()}

    def get_loss_fused(self, tb_dict):
        """Both layer losses through the one-launch kernel (same values as the tensor formulation)."""
        d = self.forward_ret_dict
        cfg = self.model_cfg.LOSS_CONFIG
        assert cfg.CLS_LOSS == 'BinaryCrossEntropy' and cfg.REG_LOSS == 'smooth-l1'
        lw = cfg.LOSS_WEIGHTS
        corner = bool(cfg.CORNER_LOSS_REGULARIZATION)
        meta = ([float(lw['rcnn_cls_weight']), float(lw['rcnn_reg_weight']),
                 float(lw['rcnn_corner_weight']) if corner else 0.0],
                [float(v) for v in lw['code_weights']], float(self.reg_loss_func.beta), corner)
        gtc = d['gt_of_rois'].shape[-1]
        parts = _FusedRcnnLoss.apply(
            d['rcnn_cls'].contiguous(), d['rcnn_reg'].contiguous().view(-1, 7),
            d['rois'].contiguous().view(-1, 7), d['gt_of_rois'].contiguous().view(-1, gtc),
            d['gt_of_rois_src'].contiguous().view(-1, gtc), d['reg_valid_mask'].contiguous().view(-1),
            d['rcnn_cls_labels'].float().contiguous().view(-1), meta)
        loss_reg = parts[1] + parts[2] if corner else parts[1]
        tb_dict.update({'rcnn_loss_cls': parts[0].detach(), 'rcnn_loss_reg': parts[1].detach()})
        if corner:
            tb_dict['rcnn_loss_corner'] = parts[2].detach()
        rcnn_loss = parts[0] + loss_reg
        tb_dict['rcnn_loss'] = rcnn_loss.detach()
        return rcnn_loss, tb_dict

    def get_loss(self, tb_dict=None, fused=None):
        tb_dict = {} if tb_dict is None else tb_dict
        d = self.forward_ret_dict
        if fused_on(fused) and d['rcnn_reg'].is_cuda and self.box_coder.code_size == 7 and d['rcnn_cls'].shape[-1] == 1 \
                and d['reg_valid_mask'].dtype == torch.int64:
            return self.get_loss_fused(tb_dict)
        loss_cls, cls_tb = self.get_box_cls_layer_loss(self.forward_ret_dict)
        loss_reg, reg_tb = self.get_box_reg_layer_loss(self.forward_ret_dict)
        tb_dict.update(cls_tb)
        tb_dict.update(reg_tb)
        rcnn_loss = loss_cls + loss_reg
        tb_dict['rcnn_loss'] = rcnn_loss.detach()
        return rcnn_loss, tb_dict
