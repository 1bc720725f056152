"""mmdet (v2.14.0) box transforms, IoU, losses and match costs that the DetMatch matching /
consistency path calls, plus the reference's own ModHungarianAssigner and
DoubleSidedFocalLossCost.

mmdet is a third-party, un-vendored dependency of the reference (README.md:15): its source is
not under /root/reference, so the formulas below restate its published algorithms
(SURVEY.md §8a-G lists them) and are "parity unpinned"; ModHungarianAssigner.assign
(mmdet3d/core/bbox/assigners/modified_hungarian_assigner.py:52-162) and
DoubleSidedFocalLossCost (mmdet3d/core/bbox/match_costs/modified_match_cost.py:9-32) are the
reference's own code and are pinned by tests/golden/hungarian.npz.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..devconst import const
from .registry import BBOX_ASSIGNERS, LOSSES, MATCH_COST, build_match_cost


# ------------------------------------------------------------------ box transforms / IoU
def bbox_xyxy_to_cxcywh(bbox):
    x1, y1, x2, y2 = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, (x2 - x1), (y2 - y1)], dim=-1)


def bbox_cxcywh_to_xyxy(bbox):
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(cx - 0.5 * w), (cy - 0.5 * h), (cx + 0.5 * w), (cy + 0.5 * h)], dim=-1)


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    """mmdet.core.bbox.iou_calculators.bbox_overlaps for xyxy boxes ('iou' | 'giou')."""
    assert mode in ('iou', 'giou')
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new_zeros((rows,) if is_aligned else (rows, cols))
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    if is_aligned:
        lt = torch.max(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.min(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap
        if mode == 'giou':
            e_lt = torch.min(bboxes1[..., :2], bboxes2[..., :2])
            e_rb = torch.max(bboxes1[..., 2:], bboxes2[..., 2:])
    else:
        lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap
        if mode == 'giou':
            e_lt = torch.min(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
            e_rb = torch.max(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    union = torch.clamp(union, min=eps)
    ious = overlap / union
    if mode == 'iou':
        return ious
    e_wh = (e_rb - e_lt).clamp(min=0)
    enclose = torch.clamp(e_wh[..., 0] * e_wh[..., 1], min=eps)
    return ious - (enclose - union) / enclose


# ------------------------------------------------------------------ losses
def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


@LOSSES.register_module()
class FocalLoss(nn.Module):
    """mmdet FocalLoss (sigmoid): pt = (1-p) t + p (1-t); w = (a t + (1-a)(1-t)) pt^g;
    loss = BCEWithLogits * w on one-hot targets, mean over n*C elements."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.use_sigmoid = use_sigmoid
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        num_classes = pred.size(1)
        t = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes].type_as(pred)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw
        if weight is not None and weight.dim() == 1:
            weight = weight.view(-1, 1)
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        if target.numel() == 0:
            return pred.sum() * 0
        return self.loss_weight * _reduce(torch.abs(pred - target), weight, reduction, avg_factor)


@LOSSES.register_module()
class MSELoss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * _reduce(F.mse_loss(pred, target, reduction='none'), weight,
                                          reduction, avg_factor)


@LOSSES.register_module()
class GIoULoss(nn.Module):

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        loss = 1 - bbox_overlaps(pred, target, mode='giou', is_aligned=True, eps=self.eps)
        return self.loss_weight * _reduce(loss, weight, reduction, avg_factor)


# ------------------------------------------------------------------ match costs
@MATCH_COST.register_module()
class FocalLossCost(object):
    """mmdet FocalLossCost: (pos - neg)[:, labels] * weight."""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class BBoxL1Cost(object):
    """mmdet BBoxL1Cost: pred (cx,cy,w,h normalised) -> xyxy when box_format == 'xyxy';
    L1 distance summed over the 4 coordinates."""

    def __init__(self, weight=1., box_format='xyxy'):
        assert box_format in ('xyxy', 'xywh')
        self.weight, self.box_format = weight, box_format

    def __call__(self, bbox_pred, gt_bboxes):
        if self.box_format == 'xywh':
            gt_bboxes = bbox_xyxy_to_cxcywh(gt_bboxes)
        else:
            bbox_pred = bbox_cxcywh_to_xyxy(bbox_pred)
        return torch.cdist(bbox_pred, gt_bboxes, p=1) * self.weight


@MATCH_COST.register_module()
class IoUCost(object):

    def __init__(self, iou_mode='giou', weight=1.):
        self.weight, self.iou_mode = weight, iou_mode

    def __call__(self, bboxes, gt_bboxes):
        return -bbox_overlaps(bboxes, gt_bboxes, mode=self.iou_mode, is_aligned=False) * self.weight


@MATCH_COST.register_module()
class DoubleSidedFocalLossCost(object):
    """modified_match_cost.py:9-32: (FL(p1, argmax p2) + FL(p2, argmax p1)^T) / 2."""

    def __init__(self, **kwargs):
        self.focal_loss_cost = FocalLossCost(**kwargs)

    def __call__(self, cls_pred_1, cls_pred_2):
        assert cls_pred_1.shape[1] == cls_pred_2.shape[1]
        cls_label_1 = cls_pred_1.sigmoid().argmax(dim=1)
        cls_label_2 = cls_pred_2.sigmoid().argmax(dim=1)
        return (self.focal_loss_cost(cls_pred_1, cls_label_2) +
                self.focal_loss_cost(cls_pred_2, cls_label_1).t()) / 2


# ------------------------------------------------------------------ assigner
class AssignResult(object):
    """mmdet AssignResult: plain record."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


def linear_sum_assignment(cost):
    """Host LAP on an (n, m) cost tensor (any device): one D2H copy of n*m floats, then
    dm_lap_host (libdetmatch_hip.so) — the reference does cost.detach().cpu() + scipy."""
    return linear_sum_assignment_host(cost.detach().cpu().numpy())


def linear_sum_assignment_host(cost_host):
    c = np.ascontiguousarray(cost_host, dtype=np.float32)
    n, m = c.shape
    k = min(n, m)
    rows = np.zeros((max(k, 1),), np.int32)
    cols = np.zeros((max(k, 1),), np.int32)
    got = _lib.lib().dm_lap_host(c.ctypes.data_as(_lib.c_f32_p), n, m,
                                 rows.ctypes.data_as(_lib.c_int_p), cols.ctypes.data_as(_lib.c_int_p))
    if got < 0:
        raise ValueError('cost matrix is infeasible (NaN)')
    return rows[:got].astype(np.int64), cols[:got].astype(np.int64)


@BBOX_ASSIGNERS.register_module()
class ModHungarianAssigner(object):
    """modified_hungarian_assigner.py:19-162: cost = cls + L1 + GIoU, one-to-one LAP;
    gt_inds 1-based (0 = unmatched); max_overlaps carries the MATCHED COST (Inf elsewhere)."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.),
                 reg_cost=dict(type='BBoxL1Cost', weight=1.0),
                 iou_cost=dict(type='IoUCost', iou_mode='giou', weight=1.0)):
        self.cls_cost = build_match_cost(cls_cost)
        self.reg_cost = build_match_cost(reg_cost)
        self.iou_cost = build_match_cost(iou_cost)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None,
               eps=1e-7):
        assert gt_bboxes_ignore is None
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        assigned_gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        assigned_labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            return AssignResult(num_gts, assigned_gt_inds, None, labels=assigned_labels)
        img_h, img_w, _ = img_meta['img_shape']
        factor = const([img_w, img_h, img_w, img_h], gt_bboxes.device, gt_bboxes.dtype).unsqueeze(0)
        cls_cost = self.cls_cost(cls_pred, gt_labels)
        reg_cost = self.reg_cost(bbox_pred, gt_bboxes / factor)
        bboxes = bbox_cxcywh_to_xyxy(bbox_pred) * factor
        iou_cost = self.iou_cost(bboxes, gt_bboxes)
        cost = (cls_cost + reg_cost + iou_cost).detach()
        rows, cols = linear_sum_assignment(cost)
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        assigned_gt_inds[:] = 0
        assigned_gt_inds[rows] = cols + 1
        if len(gt_labels.shape) == 1:
            assigned_labels[rows] = gt_labels[cols]
        max_overlaps = bbox_pred.new_full((num_bboxes,), float('Inf'))
        max_overlaps[rows] = cost[rows, cols]
        res = AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=assigned_labels)
        res.iou_cost = bbox_pred.new_full((num_bboxes,), float('Inf'))
        res.iou_cost[rows] = iou_cost.detach()[rows, cols] / self.iou_cost.weight
        return res
