"""TEST INFRASTRUCTURE — an independent restatement of the mmdet 2.14.0 / mmcv-full 1.3.16 algorithms
the DetMatch 2D branch and its matching costs call.  Only tests/, the golden generators
under tests/golden/ and __graft_entry__.smoke() may import it; the product (detmatch_amd/) never does.

PARITY UNPINNED: mmdet / mmcv are un-vendored third-party dependencies of the reference (requirements:
mmdet==2.14.0, mmcv-full==1.3.16), absent from /root/reference and from this image, so nothing here can
be run against the real packages.  Every function restates the PUBLISHED algorithm of the file it names,
sequentially and per image the way mmdet runs it (numpy for the integer / target logic, torch float64 on
the CPU where a loss gradient is wanted), and shares no code with detmatch_amd/: the product's dense,
batched, fixed-size kernels are compared against this per-image, variable-length formulation.

Call sites in the reference that reach these algorithms:
  mmdet3d/models/ssl_modules/consumers/consumers_2d.py:30,104   detector.forward_train (RPN + RoI head losses)
  mmdet3d/models/ssl_modules/processors/processors_2d.py:39-77  simple_test_rpn, bbox_roi_extractor, coder.decode
  mmdet3d/core/bbox/assigners/modified_hungarian_assigner.py:117-125   cls / reg / iou match costs
  mmdet3d/models/ssl_modules/consumers/consumers_3d.py:84-99    FocalLoss / L1Loss / GIoULoss

Random sampling: mmdet's RandomSampler.random_choice draws torch.randperm over the candidate set.  The
product draws one uniform key per box and keeps the `num` smallest keys of the candidate set — also a
uniform random subset.  `random_choice` below takes the SAME keys, so both sides select the same set.
"""
import math

import numpy as np

F32 = np.float32


# ---------------------------------------------------------------------------------------------
# mmdet/core/bbox/iou_calculators/iou2d_calculator.py: bbox_overlaps (is_aligned=False)
def bbox_overlaps(b1, b2, mode='iou', eps=1e-6):
    b1, b2 = np.asarray(b1, F32).reshape(-1, 4), np.asarray(b2, F32).reshape(-1, 4)
    out = np.zeros((len(b1), len(b2)), F32)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    for i in range(len(b1)):
        lt = np.maximum(b1[i, :2], b2[:, :2])
        rb = np.minimum(b1[i, 2:], b2[:, 2:])
        wh = np.clip(rb - lt, 0, None)
        overlap = wh[:, 0] * wh[:, 1]
        union = np.maximum(a1[i] + a2 - overlap, F32(eps))
        ious = overlap / union
        if mode == 'giou':
            elt = np.minimum(b1[i, :2], b2[:, :2])
            erb = np.maximum(b1[i, 2:], b2[:, 2:])
            ewh = np.clip(erb - elt, 0, None)
            earea = np.maximum(ewh[:, 0] * ewh[:, 1], F32(eps))
            ious = ious - (earea - union) / earea
        out[i] = ious
    return out


# mmdet/core/bbox/transforms.py
def bbox_cxcywh_to_xyxy(b):
    b = np.asarray(b, F32)
    cx, cy, w, h = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([cx - F32(0.5) * w, cy - F32(0.5) * h, cx + F32(0.5) * w, cy + F32(0.5) * h], -1)


def bbox_xyxy_to_cxcywh(b):
    b = np.asarray(b, F32)
    x1, y1, x2, y2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([(x1 + x2) / F32(2), (y1 + y2) / F32(2), x2 - x1, y2 - y1], -1)


# ---------------------------------------------------------------------------------------------
# mmdet/core/anchor/anchor_generator.py: AnchorGenerator (scale_major=True, center_offset=0)
def base_anchors(base_size, scales, ratios):
    w = h = F32(base_size)
    h_ratios = np.sqrt(np.asarray(ratios, F32))
    w_ratios = F32(1) / h_ratios
    scales = np.asarray(scales, F32)
    ws = (w * w_ratios[:, None] * scales[None, :]).reshape(-1)
    hs = (h * h_ratios[:, None] * scales[None, :]).reshape(-1)
    return np.stack([F32(-0.5) * ws, F32(-0.5) * hs, F32(0.5) * ws, F32(0.5) * hs], -1).astype(F32)


def grid_anchors(featmap_sizes, strides, scales, ratios):
    """-> list per level of (H*W*A, 4), anchor order (row, column, base anchor)."""
    out = []
    for (fh, fw), s in zip(featmap_sizes, strides):
        base = base_anchors(s, scales, ratios)
        rows = []
        for y in range(int(fh)):
            for x in range(int(fw)):
                rows.append(base + np.array([x * s, y * s, x * s, y * s], F32))
        out.append(np.concatenate(rows).astype(F32) if rows else np.zeros((0, 4), F32))
    return out


# ---------------------------------------------------------------------------------------------
# mmdet/core/bbox/coder/delta_xywh_bbox_coder.py: bbox2delta / delta2bbox
def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    p, g = np.asarray(proposals, F32), np.asarray(gt, F32)
    px, py = (p[..., 0] + p[..., 2]) * F32(0.5), (p[..., 1] + p[..., 3]) * F32(0.5)
    pw, ph = p[..., 2] - p[..., 0], p[..., 3] - p[..., 1]
    gx, gy = (g[..., 0] + g[..., 2]) * F32(0.5), (g[..., 1] + g[..., 3]) * F32(0.5)
    gw, gh = g[..., 2] - g[..., 0], g[..., 3] - g[..., 1]
    d = np.stack([(gx - px) / pw, (gy - py) / ph, np.log(gw / pw), np.log(gh / ph)], -1).astype(F32)
    return ((d - np.asarray(means, F32)) / np.asarray(stds, F32)).astype(F32)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True):
    rois, deltas = np.asarray(rois, F32), np.asarray(deltas, F32)
    n = deltas.shape[-1] // 4
    d = deltas * np.tile(np.asarray(stds, F32), n) + np.tile(np.asarray(means, F32), n)
    dx, dy, dw, dh = d[..., 0::4], d[..., 1::4], d[..., 2::4], d[..., 3::4]
    px = ((rois[..., 0] + rois[..., 2]) * F32(0.5))[..., None]
    py = ((rois[..., 1] + rois[..., 3]) * F32(0.5))[..., None]
    pw = (rois[..., 2] - rois[..., 0])[..., None]
    ph = (rois[..., 3] - rois[..., 1])[..., None]
    max_ratio = F32(abs(math.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -max_ratio, max_ratio), np.clip(dh, -max_ratio, max_ratio)
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * F32(0.5), gy - gh * F32(0.5), gx + gw * F32(0.5), gy + gh * F32(0.5)
    if clip_border and max_shape is not None:
        x1, x2 = np.clip(x1, 0, max_shape[1]), np.clip(x2, 0, max_shape[1])
        y1, y2 = np.clip(y1, 0, max_shape[0]), np.clip(y2, 0, max_shape[0])
    return np.stack([x1, y1, x2, y2], -1).reshape(deltas.shape).astype(F32)


# ---------------------------------------------------------------------------------------------
# mmdet/core/bbox/assigners/max_iou_assigner.py: MaxIoUAssigner.assign + assign_wrt_overlaps
def max_iou_assign(bboxes, gt_bboxes, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True,
                   gt_max_assign_all=True):
    """-> assigned_gt_inds (N,) int64: -1 ignore, 0 negative, k > 0: GT k-1; max_overlaps (N,)."""
    bboxes = np.asarray(bboxes, F32)[:, :4]
    gt = np.asarray(gt_bboxes, F32).reshape(-1, 4)
    n, k = len(bboxes), len(gt)
    assigned = np.full(n, -1, np.int64)
    if k == 0 or n == 0:
        if k == 0:
            assigned[:] = 0
        return assigned, np.zeros(n, F32)
    overlaps = bbox_overlaps(gt, bboxes)                          # (K, N)
    max_overlaps, argmax_overlaps = overlaps.max(0), overlaps.argmax(0)
    gt_max_overlaps, gt_argmax_overlaps = overlaps.max(1), overlaps.argmax(1)
    assigned[(max_overlaps >= 0) & (max_overlaps < neg_iou_thr)] = 0
    pos = max_overlaps >= pos_iou_thr
    assigned[pos] = argmax_overlaps[pos] + 1
    if match_low_quality:
        for i in range(k):                                        # later GTs overwrite earlier ones
            if gt_max_overlaps[i] >= min_pos_iou:
                if gt_max_assign_all:
                    assigned[overlaps[i] == gt_max_overlaps[i]] = i + 1
                else:
                    assigned[gt_argmax_overlaps[i]] = i + 1
    return assigned, max_overlaps


# mmdet/core/bbox/samplers/{base_sampler,random_sampler}.py
def random_choice(gallery, num, keys):
    """RandomSampler.random_choice with the shared-key convention of this file's header: the `num`
    gallery members with the smallest keys (ties by index)."""
    gallery = np.asarray(gallery, np.int64)
    order = np.lexsort((gallery, np.asarray(keys)[gallery]))
    return gallery[order[:num]]


def random_sample(assigned_gt_inds, bboxes, gt_bboxes, gt_labels, num, pos_fraction, add_gt_as_proposals,
                  keys):
    """BaseSampler.sample (neg_pos_ub = -1).  `keys[j]` belongs to row j of the (GT-prefixed) box list.
    -> dict(pos_inds, neg_inds (sorted, as .unique() leaves them), bboxes, gt_inds (1-based), labels)."""
    bboxes = np.asarray(bboxes, F32)[:, :4]
    gt = np.asarray(gt_bboxes, F32).reshape(-1, 4)
    assigned = np.asarray(assigned_gt_inds, np.int64)
    if add_gt_as_proposals and len(gt) > 0:
        bboxes = np.concatenate([gt, bboxes])
        assigned = np.concatenate([np.arange(1, len(gt) + 1), assigned])      # AssignResult.add_gt_
    num_expected_pos = int(num * pos_fraction)
    pos = np.nonzero(assigned > 0)[0]
    if len(pos) > num_expected_pos:
        pos = random_choice(pos, num_expected_pos, keys)
    pos = np.unique(pos)
    num_expected_neg = num - len(pos)
    neg = np.nonzero(assigned == 0)[0]
    if len(neg) > num_expected_neg:
        neg = random_choice(neg, num_expected_neg, keys)
    neg = np.unique(neg)
    return dict(pos_inds=pos, neg_inds=neg, bboxes=bboxes, assigned=assigned,
                pos_bboxes=bboxes[pos], neg_bboxes=bboxes[neg],
                pos_gt_bboxes=gt[assigned[pos] - 1] if len(gt) else np.zeros((0, 4), F32),
                pos_gt_labels=(np.asarray(gt_labels, np.int64)[assigned[pos] - 1]
                               if gt_labels is not None and len(gt) else np.zeros(0, np.int64)))


# ---------------------------------------------------------------------------------------------
# mmcv/ops/nms.py: nms (offset 0) and batched_nms
def nms(boxes, scores, iou_threshold):
    """-> keep indices in descending score order (ties: lower index first)."""
    boxes, scores = np.asarray(boxes, F32), np.asarray(scores, F32)
    order = np.argsort(-scores, kind='stable')
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    suppressed = np.zeros(len(boxes), bool)
    keep = []
    for oi, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        lt = np.maximum(boxes[i, :2], boxes[rest, :2])
        rb = np.minimum(boxes[i, 2:], boxes[rest, 2:])
        wh = np.clip(rb - lt, 0, None)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[iou > iou_threshold]] = True
    return np.asarray(keep, np.int64)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """boxes of different idxs never suppress each other: shifted by idxs * (max coordinate + 1)."""
    boxes = np.asarray(boxes, F32)
    if len(boxes) == 0:
        return np.zeros((0, 5), F32), np.zeros(0, np.int64)
    offsets = np.asarray(idxs, F32) * (boxes.max() + F32(1))
    keep = nms(boxes + offsets[:, None], scores, iou_threshold)
    return np.concatenate([boxes[keep], np.asarray(scores, F32)[keep, None]], 1), keep


# mmdet/models/dense_heads/rpn_head.py: RPNHead._get_bboxes(_single) for ONE image
def rpn_get_bboxes_single(cls_scores, bbox_preds, mlvl_anchors, img_shape, nms_pre, max_per_img, nms_thr,
                          min_bbox_size=0, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """cls_scores[l]: (A, H, W) logits, bbox_preds[l]: (4A, H, W) -> proposals (<= max_per_img, 5)."""
    lvl_scores, lvl_deltas, lvl_anchors, lvl_ids = [], [], [], []
    for lvl, (c, r, a) in enumerate(zip(cls_scores, bbox_preds, mlvl_anchors)):
        c, r = np.asarray(c, F32), np.asarray(r, F32)
        s = F32(1) / (F32(1) + np.exp(-c.transpose(1, 2, 0).reshape(-1)))
        d = r.transpose(1, 2, 0).reshape(-1, 4)
        if nms_pre > 0 and len(s) > nms_pre:
            top = np.argsort(-s, kind='stable')[:nms_pre]
            s, d, a = s[top], d[top], a[top]
        lvl_scores.append(s), lvl_deltas.append(d), lvl_anchors.append(a)
        lvl_ids.append(np.full(len(s), lvl, np.int64))
    s, d, a, ids = (np.concatenate(lvl_scores), np.concatenate(lvl_deltas), np.concatenate(lvl_anchors),
                    np.concatenate(lvl_ids))
    props = delta2bbox(a, d, means, stds, max_shape=img_shape)
    if min_bbox_size >= 0:
        ok = ((props[:, 2] - props[:, 0]) > min_bbox_size) & ((props[:, 3] - props[:, 1]) > min_bbox_size)
        props, s, ids = props[ok], s[ok], ids[ok]
    dets, _ = batched_nms(props, s, ids, nms_thr)
    return dets[:max_per_img]


# ---------------------------------------------------------------------------------------------
# mmdet/models/dense_heads/anchor_head.py: AnchorHead._get_targets_single / get_targets / loss (RPN)
def rpn_targets_single(anchors, gt_bboxes, assigner, sampler, keys, means=(0., 0., 0., 0.),
                       stds=(1., 1., 1., 1.)):
    """allowed_border = -1 and every anchor valid (the padded batch shape covers the feature maps).
    -> labels (1 = background, 0 = foreground of the one-class RPN), label_weights, bbox_targets, bbox_weights,
    pos_inds, neg_inds."""
    assigned, _ = max_iou_assign(anchors, gt_bboxes, assigner['pos_iou_thr'], assigner['neg_iou_thr'],
                                 assigner['min_pos_iou'], assigner.get('match_low_quality', True))
    s = random_sample(assigned, anchors, gt_bboxes, None, sampler['num'], sampler['pos_fraction'],
                      sampler.get('add_gt_as_proposals', False), keys)
    n = len(anchors)
    labels = np.full(n, 1, np.int64)
    label_weights = np.zeros(n, F32)
    bbox_targets, bbox_weights = np.zeros((n, 4), F32), np.zeros((n, 4), F32)
    pos, neg = s['pos_inds'], s['neg_inds']
    if len(pos):
        bbox_targets[pos] = bbox2delta(s['pos_bboxes'], s['pos_gt_bboxes'], means, stds)
        bbox_weights[pos] = 1.0
        labels[pos] = 0
        label_weights[pos] = 1.0
    if len(neg):
        label_weights[neg] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights, pos, neg


def rpn_loss(cls_scores, bbox_preds, anchors, gt_bboxes_list, assigner, sampler, keys, loss_cls_weight=1.0,
             loss_bbox_weight=1.0):
    """AnchorHead.loss for the RPN (CrossEntropyLoss use_sigmoid, L1Loss, sampling): the per-level
    loss lists summed.  cls_scores[l]: torch (B, A, H, W), bbox_preds[l]: (B, 4A, H, W), float64 leaves or
    not; anchors: (N, 4) numpy, all levels concatenated; keys: (B, N).
    num_total_samples = sum_i max(#pos_i, 1) + sum_i max(#neg_i, 1)   (anchor_head.py get_targets).
    -> (loss_rpn_cls, loss_rpn_bbox) torch float64 scalars (differentiable), per-image target tuples."""
    import torch
    import torch.nn.functional as Fn
    b = cls_scores[0].shape[0]
    score = torch.cat([c.permute(0, 2, 3, 1).reshape(b, -1) for c in cls_scores], 1).double()
    delta = torch.cat([r.permute(0, 2, 3, 1).reshape(b, -1, 4) for r in bbox_preds], 1).double()
    targets, num_pos, num_neg = [], 0, 0
    for i in range(b):
        t = rpn_targets_single(anchors, gt_bboxes_list[i], assigner, sampler, keys[i])
        targets.append(t)
        num_pos += max(len(t[4]), 1)
        num_neg += max(len(t[5]), 1)
    num_total = float(num_pos + num_neg)
    loss_cls = score.new_zeros(())
    loss_bbox = score.new_zeros(())
    for i, (labels, lw, bt, bw, _, _) in enumerate(targets):
        target = torch.from_numpy((labels == 0).astype(np.float64))           # _expand_onehot_labels
        bce = Fn.binary_cross_entropy_with_logits(score[i], target, reduction='none')
        loss_cls = loss_cls + (bce * torch.from_numpy(lw.astype(np.float64))).sum()
        l1 = (delta[i] - torch.from_numpy(bt.astype(np.float64))).abs()
        loss_bbox = loss_bbox + (l1 * torch.from_numpy(bw.astype(np.float64))).sum()
    return loss_cls_weight * loss_cls / num_total, loss_bbox_weight * loss_bbox / num_total, targets


# ---------------------------------------------------------------------------------------------
# mmdet/models/roi_heads/standard_roi_head.py forward_train (assign + sample per image) and
# mmdet/models/roi_heads/bbox_heads/bbox_head.py: _get_target_single / get_targets
def roi_targets_single(proposals, gt_bboxes, gt_labels, assigner, sampler, keys, num_classes,
                       means=(0., 0., 0., 0.), stds=(0.1, 0.1, 0.2, 0.2)):
    """proposals: the image's VALID proposals (n, >= 4); keys: one per row of [GT..., proposals...] when
    add_gt_as_proposals (and GT exist), else one per proposal.
    -> rois (S, 4) [positives then negatives, index order], labels, label_weights, bbox_targets, bbox_weights."""
    assigned, _ = max_iou_assign(proposals, gt_bboxes, assigner['pos_iou_thr'], assigner['neg_iou_thr'],
                                 assigner['min_pos_iou'], assigner.get('match_low_quality', False))
    s = random_sample(assigned, proposals, gt_bboxes, gt_labels, sampler['num'], sampler['pos_fraction'],
                      sampler.get('add_gt_as_proposals', True), keys)
    num_pos, num_neg = len(s['pos_inds']), len(s['neg_inds'])
    n = num_pos + num_neg
    labels = np.full(n, num_classes, np.int64)
    label_weights = np.zeros(n, F32)
    bbox_targets, bbox_weights = np.zeros((n, 4), F32), np.zeros((n, 4), F32)
    if num_pos:
        labels[:num_pos] = s['pos_gt_labels']
        label_weights[:num_pos] = 1.0
        bbox_targets[:num_pos] = bbox2delta(s['pos_bboxes'], s['pos_gt_bboxes'], means, stds)
        bbox_weights[:num_pos] = 1.0
    if num_neg:
        label_weights[-num_neg:] = 1.0
    rois = np.concatenate([s['pos_bboxes'], s['neg_bboxes']]).astype(F32)
    return rois, labels, label_weights, bbox_targets, bbox_weights


# mmdet/models/losses/focal_loss.py (sigmoid_focal_loss / py_sigmoid_focal_loss), utils.py weight_reduce_loss
def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, avg_factor=None):
    """pred (N, C) torch logits; target (N,) long in [0, C] (C = no positive channel) -> 'mean' reduction."""
    import torch
    import torch.nn.functional as Fn
    pred = pred.double()
    onehot = Fn.one_hot(target.long(), pred.shape[1] + 1)[:, :pred.shape[1]].double()
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    fw = (alpha * onehot + (1 - alpha) * (1 - onehot)) * pt.pow(gamma)
    loss = Fn.binary_cross_entropy_with_logits(pred, onehot, reduction='none') * fw
    if weight is not None:
        loss = loss * weight.double().view(-1, 1)
    return loss.sum() / avg_factor if avg_factor is not None else loss.mean()


# mmdet/models/roi_heads/bbox_heads/bbox_head.py: BBoxHead.loss (FocalLoss cls, L1Loss box, class-specific)
def bbox_head_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_classes,
                   alpha=0.5, gamma=2.0, loss_cls_weight=1.0, loss_bbox_weight=1.0):
    """torch tensors in -> (loss_cls, loss_bbox, acc) float64 (differentiable w.r.t. cls_score / bbox_pred)."""
    import torch
    avg_factor = max(float((label_weights > 0).sum()), 1.0)
    loss_cls = loss_cls_weight * sigmoid_focal_loss(cls_score, labels, label_weights, gamma, alpha, avg_factor)
    pred_label = cls_score.argmax(1)                                        # accuracy(): top-1 over all rows
    acc = (pred_label == labels).double().sum() * 100.0 / max(cls_score.shape[0], 1)
    pos = (labels >= 0) & (labels < num_classes)
    if bool(pos.any()):
        pb = bbox_pred.double().view(bbox_pred.shape[0], -1, 4)[pos, labels[pos]]
        l1 = (pb - bbox_targets[pos].double()).abs() * bbox_weights[pos].double()
        loss_bbox = loss_bbox_weight * l1.sum() / float(bbox_targets.shape[0])
    else:
        loss_bbox = bbox_pred.double()[pos].sum()
    return loss_cls, loss_bbox, acc


# mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py: map_roi_levels
def map_roi_levels(rois, num_levels, finest_scale=56):
    rois = np.asarray(rois, F32)
    scale = np.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = np.floor(np.log2(scale / F32(finest_scale) + F32(1e-6)))
    return np.clip(lv, 0, num_levels - 1).astype(np.int64)


# ---------------------------------------------------------------------------------------------
# mmdet/core/bbox/match_costs/match_cost.py
def focal_loss_cost(cls_pred, gt_labels, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
    """FocalLossCost.__call__: cls_pred (N, C) logits, gt_labels (M,) -> (N, M)."""
    p = F32(1) / (F32(1) + np.exp(-np.asarray(cls_pred, F32)))
    neg = -np.log(F32(1) - p + F32(eps)) * F32(1 - alpha) * p ** gamma
    pos = -np.log(p + F32(eps)) * F32(alpha) * (F32(1) - p) ** gamma
    return ((pos[:, gt_labels] - neg[:, gt_labels]) * F32(weight)).astype(F32)


def bbox_l1_cost(bbox_pred, gt_bboxes, weight=1.0, box_format='xyxy'):
    """BBoxL1Cost.__call__: bbox_pred normalised (cx, cy, w, h), gt_bboxes normalised xyxy; torch.cdist p=1."""
    g = np.asarray(gt_bboxes, F32)
    b = np.asarray(bbox_pred, F32)
    if box_format == 'xywh':
        g = bbox_xyxy_to_cxcywh(g)
    else:
        b = bbox_cxcywh_to_xyxy(b)
    return (np.abs(b[:, None, :] - g[None, :, :]).sum(-1) * F32(weight)).astype(F32)


def iou_cost(bboxes, gt_bboxes, weight=1.0, iou_mode='giou'):
    """IoUCost.__call__: both unnormalised xyxy -> -IoU * weight."""
    return (-bbox_overlaps(bboxes, gt_bboxes, mode=iou_mode) * F32(weight)).astype(F32)


# ---------------------------------------------------------------------------------------------
# mmdet/models/losses: FocalLoss (sigmoid, class-index targets), L1Loss, GIoULoss — 'mean' reduction
def focal_loss_mean(pred_logits, target, alpha=0.25, gamma=2.0, loss_weight=1.0):
    return loss_weight * sigmoid_focal_loss(pred_logits, target, None, gamma, alpha, None)


def l1_loss_mean(pred, target, loss_weight=1.0):
    import torch
    if target.numel() == 0:
        return pred.sum() * 0
    return loss_weight * (pred.double() - target.double()).abs().mean()


def giou_loss_mean(pred, target, eps=1e-6, loss_weight=1.0):
    """iou_loss.py giou_loss: 1 - bbox_overlaps(pred, target, 'giou', is_aligned=True, eps)."""
    import torch
    p, t = pred.double(), target.double()
    if p.numel() == 0:
        return p.sum() * 0
    a1 = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    a2 = (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    lt, rb = torch.max(p[:, :2], t[:, :2]), torch.min(p[:, 2:], t[:, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    union = (a1 + a2 - overlap).clamp(min=eps)
    ious = overlap / union
    elt, erb = torch.min(p[:, :2], t[:, :2]), torch.max(p[:, 2:], t[:, 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = (ewh[:, 0] * ewh[:, 1]).clamp(min=eps)
    gious = ious - (earea - union) / earea
    return loss_weight * (1 - gious).mean()
