"""Faster R-CNN (two-stage) for the DetMatch 2D branch — mmdet 2.14.0 TwoStageDetector / RPNHead /
StandardRoIHead / Shared2FCBBoxHead / MaxIoUAssigner / RandomSampler / DeltaXYWHBBoxCoder /
AnchorGenerator restated (un-vendored third party: PARITY UNPINNED) for
configs/detmatch/001/detmatch/split_0.py:39-99 (model), :440-478 (train_cfg), :507-529 (test_cfg).

Differences in HOW (not what) it computes:
  * target assignment + random sampling are dense (masks, top-k over random keys): no nonzero(),
    no randperm over data-dependent index sets, no host synchronisation in the training step;
  * RPN proposals are fixed-size (max_per_img rows + validity mask), NMS on the device
    (dm_nms_2d), one RoIAlign launch for the whole pyramid (dm_roi_align_forward);
  * the per-level loss lists of mmdet (5 entries per RPN loss) are returned already summed — the
    value `_parse_losses` would log and back-propagate is identical.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, dense_conv
from ..devconst import const
from ..mm3d.base_detector import DetectorStepMixin
from ..mm3d.losses import bbox_overlaps
from ..mm3d.registry import DETECTORS, LOSSES, build_from_cfg
from ..fused import on as fused_on
from ..roi_align import roi_align_fpn
from .backbone import FPN, ResNet


# ------------------------------------------------------------------ anchors / coder
class AnchorGenerator(object):
    """mmdet AnchorGenerator (scale_major, center_offset 0): anchors ordered (H, W, A)."""

    def __init__(self, strides, ratios, scales, **kwargs):
        self.strides = [(s, s) if isinstance(s, int) else tuple(s) for s in strides]
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self.scales = torch.tensor(scales, dtype=torch.float32)
        self.base_anchors = [self._base(s[0]) for s in self.strides]
        self._cache = {}

    @property
    def num_base_anchors(self):
        return [b.shape[0] for b in self.base_anchors]

    def _base(self, base_size):
        w = h = float(base_size)
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        ws = (w * w_ratios[:, None] * self.scales[None, :]).view(-1)
        hs = (h * h_ratios[:, None] * self.scales[None, :]).view(-1)
        return torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], dim=-1)

    def grid_anchors(self, featmap_sizes, device):
        key = (tuple(tuple(int(v) for v in s) for s in featmap_sizes), str(device))
        if key not in self._cache:
            out = []
            for (h, w), (sw, sh), base in zip(featmap_sizes, self.strides, self.base_anchors):
                sx = torch.arange(0, w, device=device, dtype=torch.float32) * sw
                sy = torch.arange(0, h, device=device, dtype=torch.float32) * sh
                xx = sx.repeat(h)
                yy = sy.view(-1, 1).repeat(1, w).view(-1)
                shifts = torch.stack([xx, yy, xx, yy], dim=-1)
                out.append((base.to(device)[None, :, :] + shifts[:, None, :]).view(-1, 4))
            self._cache[key] = out
        return self._cache[key]


class DeltaXYWHBBoxCoder(object):

    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True,
                 **kwargs):
        self.means, self.stds, self.clip_border = list(target_means), list(target_stds), clip_border

    def encode(self, proposals, gt):
        p, g = proposals.float(), gt.float()
        px, py = (p[..., 0] + p[..., 2]) * 0.5, (p[..., 1] + p[..., 3]) * 0.5
        pw, ph = p[..., 2] - p[..., 0], p[..., 3] - p[..., 1]
        gx, gy = (g[..., 0] + g[..., 2]) * 0.5, (g[..., 1] + g[..., 3]) * 0.5
        gw, gh = g[..., 2] - g[..., 0], g[..., 3] - g[..., 1]
        d = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
        return (d - const(self.means, d.device, d.dtype)) / const(self.stds, d.device, d.dtype)

    def decode(self, rois, deltas, max_shape=None, wh_ratio_clip=16 / 1000):
        n = deltas.size(-1) // 4
        means = const(self.means * n, deltas.device, deltas.dtype)
        stds = const(self.stds * n, deltas.device, deltas.dtype)
        d = deltas * stds + means
        dx, dy, dw, dh = d[..., 0::4], d[..., 1::4], d[..., 2::4], d[..., 3::4]
        max_ratio = abs(np.log(wh_ratio_clip))
        dw, dh = dw.clamp(-max_ratio, max_ratio), dh.clamp(-max_ratio, max_ratio)
        px = ((rois[..., 0] + rois[..., 2]) * 0.5).unsqueeze(-1)
        py = ((rois[..., 1] + rois[..., 3]) * 0.5).unsqueeze(-1)
        pw = (rois[..., 2] - rois[..., 0]).unsqueeze(-1)
        ph = (rois[..., 3] - rois[..., 1]).unsqueeze(-1)
        gw, gh = pw * dw.exp(), ph * dh.exp()
        gx, gy = px + pw * dx, py + ph * dy
        x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
        if self.clip_border and max_shape is not None:
            x1, x2 = x1.clamp(0, max_shape[1]), x2.clamp(0, max_shape[1])
            y1, y2 = y1.clamp(0, max_shape[0]), y2.clamp(0, max_shape[0])
        return torch.stack([x1, y1, x2, y2], dim=-1).view(deltas.size())


# ------------------------------------------------------------------ assigner / sampler (dense)
def max_iou_assign(boxes, gt_boxes, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality,
                   box_valid=None):
    """mmdet MaxIoUAssigner.assign_wrt_overlaps (gt_max_assign_all True, no ignore):
    -> assigned_gt_inds (N) long: -1 ignore, 0 negative, k>0 = GT k (1-based)."""
    n, k = boxes.shape[0], gt_boxes.shape[0]
    if k == 0:
        assigned = boxes.new_zeros((n,), dtype=torch.long)
    else:
        ov = bbox_overlaps(gt_boxes, boxes)                       # (K, N)
        max_ov, argmax_ov = ov.max(dim=0)
        assigned = boxes.new_full((n,), -1, dtype=torch.long)
        assigned = torch.where((max_ov >= 0) & (max_ov < neg_iou_thr), torch.zeros_like(assigned), assigned)
        assigned = torch.where(max_ov >= pos_iou_thr, argmax_ov + 1, assigned)
        if match_low_quality:
            gt_max = ov.max(dim=1)[0]
            hit = (ov == gt_max[:, None]) & (gt_max >= min_pos_iou)[:, None]
            # the reference loops over GTs in order, later GTs overwrite: take the largest index
            gid = torch.arange(1, k + 1, device=boxes.device)[:, None]
            low = (hit.long() * gid).max(dim=0)[0]
            assigned = torch.where(low > 0, low, assigned)
    if box_valid is not None:
        assigned = torch.where(box_valid, assigned, torch.full_like(assigned, -1))
    return assigned


def random_sample(assigned, num, pos_fraction, generator=None, keys=None):
    """mmdet RandomSampler (neg_pos_ub -1): up to int(num*pos_fraction) positives, the rest
    negatives, uniformly at random.  -> (pos_idx (P), pos_ok (P) bool, neg_idx (num), neg_ok).
    `keys`: one uniform number per box (drawn here when None); the boxes with the smallest keys win."""
    n = assigned.shape[0]
    n_pos = min(int(num * pos_fraction), n)
    n_neg = min(num, n)
    key = torch.rand((n,), device=assigned.device, generator=generator) if keys is None else keys[:n]
    two = key.new_full((), 2.0)
    vp, ip = torch.topk(torch.where(assigned > 0, key, two), n_pos, largest=False)
    pos_ok = vp < 1.5
    vn, ineg = torch.topk(torch.where(assigned == 0, key, two), n_neg, largest=False)
    budget = num - pos_ok.sum()
    neg_ok = (vn < 1.5) & (torch.arange(n_neg, device=assigned.device) < budget)
    return ip, pos_ok, ineg, neg_ok


def nms_fixed(boxes, scores, iou_thr, max_num):
    """Greedy NMS on the device -> (idx (max_num) into boxes, ok (max_num) bool), score order."""
    order = _lib.sort_rows(scores, descending=True)
    b = boxes[order].contiguous().float()
    _lib.require_device(b)
    n = b.shape[0]
    L = _lib.lib()
    keep = torch.zeros((max(n, max_num),), dtype=torch.int64, device=b.device)
    num = torch.zeros((1,), dtype=torch.int32, device=b.device)
    ws = _lib.workspace(L.dm_nms_workspace_bytes(n), b.device, 'nms')
    _lib.check(L.dm_nms_2d(_lib.ptr(b), n, float(iou_thr), int(max_num), _lib.ptr(keep), _lib.ptr(num),
                           _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_nms_2d')
    ok = torch.arange(max_num, device=b.device) < num
    idx = torch.where(ok, keep[:max_num], torch.zeros_like(keep[:max_num]))
    return order[idx], ok


def nms_fixed_batch(boxes, scores, iou_thr, max_num):
    """nms_fixed for B images of N candidates each in one launch chain -> (idx (B, max_num), ok (B, max_num))."""
    order = _lib.sort_rows(scores, descending=True)
    b = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4)).contiguous().float()
    _lib.require_device(b)
    bsz, n = b.shape[0], b.shape[1]
    L = _lib.lib()
    keep = torch.zeros((bsz, max(n, max_num)), dtype=torch.int64, device=b.device)
    num = torch.zeros((bsz,), dtype=torch.int32, device=b.device)
    ws = _lib.workspace(L.dm_nms_workspace_bytes(n) * bsz, b.device, 'nms')
    _lib.check(L.dm_nms_2d_batch(_lib.ptr(b), bsz, n, float(iou_thr), int(max_num), _lib.ptr(keep), keep.stride(0),
                                 _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_nms_2d_batch')
    ok = torch.arange(max_num, device=b.device)[None, :] < num[:, None]
    idx = torch.where(ok, keep[:, :max_num], torch.zeros_like(keep[:, :max_num]))
    return torch.gather(order, 1, idx), ok


# ------------------------------------------------------------------ fused targets / losses
def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * max(len(tensors), 1))(*[t.data_ptr() if t is not None and t.numel() else None
                                                     for t in tensors])


def _gt_arrays(gt_bboxes):
    gts = [g.detach().float().contiguous() for g in gt_bboxes]
    return gts, _ptr_array(gts), _lib.ints([g.shape[0] for g in gts])


class _FusedRpnLoss(torch.autograd.Function):
    """dm_rpn_loss_forward / _backward (csrc/det2d_targets.hip): assignment, sampling and both RPN
    losses of the whole batch read straight from the NHWC head outputs -> tensor [cls, bbox]; the
    gradient comes back as one scatter into a zeroed buffer that the per-level gradients are views of."""

    @staticmethod
    def forward(ctx, meta, anchors, keys, *levels):
        (gt_bboxes, a_cfg, num, n_pos, means, stds, w_cls, w_box, n_base) = meta
        L = _lib.lib()
        ys = [dense_conv._cl(y) for y in levels]
        _lib.require_device(anchors, keys)             # ys: dense NHWC (checked by _cl)
        if not all(y.is_cuda for y in ys):
            raise _lib.DetMatchHipError('the RPN loss kernel runs on the MI355X only')
        b, c = int(ys[0].shape[0]), int(ys[0].shape[1])
        hw = [int(y.shape[2] * y.shape[3]) for y in ys]
        offs, total = [], 0
        for n in hw:
            offs.append(total)
            total += b * n * c
        gts, gt_ptrs, n_gt = _gt_arrays(gt_bboxes)
        n_anchors = int(anchors.shape[0])
        dev = anchors.device
        n_ent = b * (n_pos + num) * 5
        ent_off = torch.empty(n_ent, dtype=torch.int64, device=dev)
        ent_val = torch.empty(n_ent, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        import ctypes
        ws = _lib.workspace(L.dm_det2d_assign_workspace_bytes(b, n_anchors), dev, 'det2d')
        _lib.check(L.dm_rpn_loss_forward(
            _ptr_array(ys), _lib.ints(hw), len(ys), n_base, c, (ctypes.c_longlong * len(offs))(*offs),
            _lib.ptr(anchors), n_anchors, gt_ptrs, n_gt, b, _lib.ptr(keys), float(a_cfg['pos_iou_thr']),
            float(a_cfg['neg_iou_thr']), float(a_cfg['min_pos_iou']),
            int(bool(a_cfg.get('match_low_quality', True))), num, n_pos, _lib.floats(means), _lib.floats(stds),
            float(w_cls), float(w_box), _lib.ptr(out), _lib.ptr(ent_off), _lib.ptr(ent_val), None,
            _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_rpn_loss_forward')
        ctx.save_for_backward(ent_off, ent_val)
        ctx.layout = (total, offs, [tuple(y.shape) for y in ys])
        return out

    @staticmethod
    def backward(ctx, grad):
        ent_off, ent_val = ctx.saved_tensors
        total, offs, shapes = ctx.layout
        flat = torch.zeros(total, dtype=torch.float32, device=ent_val.device)
        _lib.check(_lib.lib().dm_rpn_loss_backward(
            _lib.ptr(ent_off), _lib.ptr(ent_val), _lib.ptr(grad.contiguous().float()), int(ent_off.numel()),
            _lib.ptr(flat), _lib.stream()), 'dm_rpn_loss_backward')
        grads = [flat[o:o + b * c * h * w].view(b, h, w, c).permute(0, 3, 1, 2)
                 for o, (b, c, h, w) in zip(offs, shapes)]
        return (None, None, None) + tuple(grads)


class _FusedBBoxHeadLoss(torch.autograd.Function):
    """dm_bbox_head_loss: focal classification loss, accuracy and class-specific L1 -> [cls, bbox, acc]."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, meta):
        n_classes, agnostic, alpha, w_cls, w_box = meta
        _lib.require_device(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights)
        out = torch.empty(3, dtype=torch.float32, device=cls_score.device)
        g_cls, g_box = torch.empty_like(cls_score), torch.empty_like(bbox_pred)
        _lib.check(_lib.lib().dm_bbox_head_loss(
            _lib.ptr(cls_score), _lib.ptr(bbox_pred), _lib.ptr(labels), _lib.ptr(label_weights),
            _lib.ptr(bbox_targets), _lib.ptr(bbox_weights), int(cls_score.shape[0]), int(cls_score.shape[1]),
            int(n_classes), int(agnostic), float(alpha), float(w_cls), float(w_box), _lib.ptr(out),
            _lib.ptr(g_cls), _lib.ptr(g_box), _lib.stream()), 'dm_bbox_head_loss')
        ctx.save_for_backward(g_cls, g_box)
        return out

    @staticmethod
    def backward(ctx, grad):
        g_cls, g_box = ctx.saved_tensors
        return g_cls * grad[0], g_box * grad[1], None, None, None, None, None


# ------------------------------------------------------------------ RPN
class RPNHead(nn.Module):

    def __init__(self, in_channels, feat_channels, anchor_generator, bbox_coder, loss_cls=None,
                 loss_bbox=None, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        ag = dict(anchor_generator)
        ag.pop('type', None)
        self.anchor_generator = AnchorGenerator(**ag)
        bc = dict(bbox_coder)
        bc.pop('type', None)
        self.bbox_coder = DeltaXYWHBBoxCoder(**bc)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.loss_cls_weight = (loss_cls or {}).get('loss_weight', 1.0)
        self.loss_bbox_weight = (loss_bbox or {}).get('loss_weight', 1.0)
        assert (loss_cls or {}).get('use_sigmoid', True)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.rpn_conv = dense_conv.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)

    def forward(self, feats):
        """3x3 conv + ReLU (fused in the GEMM epilogue), then the objectness and delta 1x1 convs as ONE
        GEMM (A + 4A output channels, zero-padded to a multiple of 4), per pyramid level."""
        a = self.num_anchors
        pad = (-5 * a) % 4
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight], dim=0)
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias], dim=0)
        if pad:
            w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)
            b = torch.cat([b, b.new_zeros(pad)], dim=0)
        cls, reg, raw = [], [], []
        for x in feats:
            x = dense_conv.conv2d(x, self.rpn_conv.weight, self.rpn_conv.bias, 1, 1, relu=True)
            y = dense_conv.conv2d(x, w, b, 1, 0)
            raw.append(y)
            cls.append(y[:, :a])
            reg.append(y[:, a:5 * a])
        self._raw_levels = raw           # what the fused loss reads (objectness + deltas, NHWC)
        return cls, reg

    @staticmethod
    def _flatten(cls, reg):
        """-> scores (B, N_total) logits, deltas (B, N_total, 4), level sizes; anchor order (H,W,A)."""
        b = cls[0].shape[0]
        s = torch.cat([c.permute(0, 2, 3, 1).reshape(b, -1) for c in cls], dim=1)
        d = torch.cat([r.permute(0, 2, 3, 1).reshape(b, -1, 4) for r in reg], dim=1)
        return s, d

    def _all_anchors(self, sizes, device):
        key = (tuple(tuple(int(v) for v in s) for s in sizes), str(device))
        cache = self.__dict__.setdefault('_anchor_cat', {})
        if key not in cache:
            cache[key] = torch.cat(self.anchor_generator.grid_anchors(sizes, device), dim=0).contiguous()
        return cache[key]

    def loss(self, cls, reg, gt_bboxes, img_metas, fused=None, keys=None):
        """mmdet AnchorHead.loss with sampling: BCE over the 256 sampled anchors, L1 over encoded
        deltas of the sampled positives, both / num_total_samples = sum over the images of max(#pos, 1) +
        max(#neg, 1) (anchor_head.py get_targets)."""
        cfg = self.train_cfg
        a_cfg, s_cfg = cfg['assigner'], cfg['sampler']
        sizes = [c.shape[-2:] for c in cls]
        anchors = self._all_anchors(sizes, cls[0].device)
        if keys is None:
            keys = torch.rand((cls[0].shape[0], anchors.shape[0]), device=anchors.device)
        raw = getattr(self, '_raw_levels', None)
        if fused_on(fused) and anchors.is_cuda and raw is not None and len(raw) == len(cls) \
                and all(c.data_ptr() == y.data_ptr() and c.shape[2:] == y.shape[2:] and c.requires_grad == y.requires_grad
                        for c, y in zip(cls, raw)) and cls[0].shape[0] <= 8 \
                and max(g.shape[0] for g in gt_bboxes) <= 256:
            meta = (list(gt_bboxes), a_cfg, int(s_cfg['num']), int(s_cfg['num'] * s_cfg['pos_fraction']),
                    self.bbox_coder.means, self.bbox_coder.stds, self.loss_cls_weight, self.loss_bbox_weight,
                    self.num_anchors)
            parts = _FusedRpnLoss.apply(meta, anchors, keys, *raw)
            return dict(loss_rpn_cls=parts[0], loss_rpn_bbox=parts[1])
        scores, deltas = self._flatten(cls, reg)
        tot_cls, tot_box, tot_n = 0., 0., 0.
        for i in range(scores.shape[0]):
            gt = gt_bboxes[i]
            with torch.no_grad():
                assigned = max_iou_assign(anchors, gt, a_cfg['pos_iou_thr'], a_cfg['neg_iou_thr'],
                                          a_cfg['min_pos_iou'], a_cfg.get('match_low_quality', True))
                ip, pos_ok, ineg, neg_ok = random_sample(assigned, s_cfg['num'], s_cfg['pos_fraction'],
                                                         keys=keys[i])
                if gt.shape[0] > 0:
                    tgt = self.bbox_coder.encode(anchors[ip], gt[(assigned[ip] - 1).clamp(min=0)])
                    tgt = torch.where(pos_ok[:, None], tgt, torch.zeros_like(tgt))
                else:
                    tgt = anchors.new_zeros((ip.shape[0], 4))
            idx = torch.cat([ip, ineg])
            w = torch.cat([pos_ok, neg_ok]).float()
            label = torch.cat([torch.ones_like(pos_ok, dtype=torch.float32),
                               torch.zeros_like(neg_ok, dtype=torch.float32)])
            bce = F.binary_cross_entropy_with_logits(scores[i, idx], label, reduction='none')
            tot_cls = tot_cls + (bce * w).sum()
            tot_box = tot_box + ((deltas[i, ip] - tgt).abs() * pos_ok[:, None].float()).sum()
            # mmdet anchor_head.py get_targets: num_total_samples = sum_i max(#pos_i, 1) + sum_i max(#neg_i, 1)
            tot_n = tot_n + pos_ok.sum().clamp(min=1).float() + neg_ok.sum().clamp(min=1).float()
        return dict(loss_rpn_cls=self.loss_cls_weight * tot_cls / tot_n,
                    loss_rpn_bbox=self.loss_bbox_weight * tot_box / tot_n)

    def _pre_nms_device(self, raw, sizes, img_metas, cfg):
        """dm_rpn_proposals_pre_nms: top-k per level, decode, flags and batched-NMS inputs of every image
        in two launches -> boxes (B, T, 4), scores (B, T), live (B, T) bool, nms boxes / scores."""
        import ctypes
        L = _lib.lib()
        ys = [dense_conv._cl(y.detach()) for y in raw]
        b, c, a = int(ys[0].shape[0]), int(ys[0].shape[1]), self.num_anchors
        hw = [int(h * w) for h, w in sizes]
        nms_pre = int(cfg['nms_pre'])
        t = sum(min(nms_pre, n * a) if nms_pre > 0 else n * a for n in hw)
        anchors = self._all_anchors(sizes, ys[0].device)
        dev = anchors.device
        boxes = torch.empty((b, t, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((b, t), dtype=torch.float32, device=dev)
        live = torch.empty((b, t), dtype=torch.bool, device=dev)
        b_nms, s_nms = torch.empty_like(boxes), torch.empty_like(scores)
        img_hw = []
        for m in img_metas:
            img_hw += [float(m['img_shape'][0]), float(m['img_shape'][1])]
        # + one selection key per anchor and image (optional part of the workspace: the keys are then
        # computed once by a chip-wide pre-pass instead of four times inside the one-CU selection kernel)
        ws = _lib.workspace(L.dm_rpn_proposals_workspace_bytes(b, t) + b * int(anchors.shape[0]) * 4 + 512, dev,
                            'rpn_props')
        coder = self.bbox_coder
        _lib.check(L.dm_rpn_proposals_pre_nms(
            _ptr_array(ys), _lib.ints(hw), len(ys), a, c, _lib.ptr(anchors), int(anchors.shape[0]), b,
            _lib.floats(img_hw), nms_pre, _lib.floats(coder.means), _lib.floats(coder.stds),
            float(abs(np.log(16 / 1000))), int(bool(coder.clip_border)), float(cfg.get('min_bbox_size', 0)), t,
            _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(live), _lib.ptr(b_nms), _lib.ptr(s_nms), _lib.ptr(ws),
            ws.numel(), _lib.stream()), 'dm_rpn_proposals_pre_nms')
        return boxes, scores, live, b_nms, s_nms

    @torch.no_grad()
    def get_bboxes(self, cls, reg, img_metas, cfg, fused=None):
        """mmdet RPNHead._get_bboxes_single per image: per-level top nms_pre, decode, drop empty
        boxes, NMS across levels (boxes of different levels never suppress each other), keep
        max_per_img.  -> list of (proposals (max_per_img, 5) [x1,y1,x2,y2,score], ok (max_per_img))."""
        sizes = [c.shape[-2:] for c in cls]
        nms_thr = cfg['nms'].get('iou_threshold', cfg['nms'].get('iou_thr', 0.7))
        raw = getattr(self, '_raw_levels', None)
        if fused_on(fused) and cls[0].is_cuda and raw is not None and len(raw) == len(cls) and cls[0].shape[0] <= 8 \
                and 0 < cfg['nms_pre'] <= 2048 \
                and all(c.data_ptr() == y.data_ptr() and c.shape[2:] == y.shape[2:] for c, y in zip(cls, raw)):
            boxes, s, live, b_nms, s_nms = self._pre_nms_device(raw, sizes, img_metas, cfg)
            out = []
            same = isinstance(b_nms, torch.Tensor) and isinstance(s_nms, torch.Tensor) and b_nms.dim() == 3
            if same:       # every image has the same candidate count: all images in one launch chain
                idx_all, ok_all = nms_fixed_batch(b_nms, s_nms, nms_thr, cfg['max_per_img'])
            for i in range(cls[0].shape[0]):
                idx, ok = (idx_all[i], ok_all[i]) if same else nms_fixed(b_nms[i], s_nms[i], nms_thr, cfg['max_per_img'])
                ok = ok & live[i][idx]
                props = torch.cat([boxes[i][idx], s[i][idx, None]], dim=1) * ok[:, None].float()
                out.append((props, ok))
            return out
        mlvl_anchors = self.anchor_generator.grid_anchors(sizes, cls[0].device)
        out = []
        for i in range(cls[0].shape[0]):
            ss, dd, aa, ll = [], [], [], []
            for lvl, (c, r, a) in enumerate(zip(cls, reg, mlvl_anchors)):
                s = c[i].permute(1, 2, 0).reshape(-1).sigmoid()
                d = r[i].permute(1, 2, 0).reshape(-1, 4)
                if cfg['nms_pre'] > 0 and s.shape[0] > cfg['nms_pre']:
                    s, top = torch.topk(s, cfg['nms_pre'])
                    d, a = d[top], a[top]
                ss.append(s), dd.append(d), aa.append(a)
                ll.append(s.new_full((s.shape[0],), lvl))
            s, d, a, l = torch.cat(ss), torch.cat(dd), torch.cat(aa), torch.cat(ll)
            boxes = self.bbox_coder.decode(a, d, max_shape=img_metas[i]['img_shape'])
            live = torch.ones_like(s, dtype=torch.bool)
            if cfg.get('min_bbox_size', 0) >= 0:
                mn = cfg.get('min_bbox_size', 0)
                live = ((boxes[:, 2] - boxes[:, 0]) > mn) & ((boxes[:, 3] - boxes[:, 1]) > mn)
            # dropped boxes: pushed to the end of the score order and far away from everything
            s_nms = torch.where(live, s, s.new_full((), -1.0))
            off = l * (boxes.max() + 1)
            b_nms = torch.where(live[:, None], boxes + off[:, None], boxes.new_full((), -1e6))
            idx, ok = nms_fixed(b_nms, s_nms, nms_thr, cfg['max_per_img'])
            ok = ok & live[idx]
            props = torch.cat([boxes[idx], s[idx, None]], dim=1) * ok[:, None].float()
            out.append((props, ok))
        return out


# ------------------------------------------------------------------ RoI head
class Shared2FCBBoxHead(nn.Module):

    def __init__(self, in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                 bbox_coder=None, reg_class_agnostic=False, loss_cls=None, loss_bbox=None, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.reg_class_agnostic = reg_class_agnostic
        bc = dict(bbox_coder)
        bc.pop('type', None)
        self.bbox_coder = DeltaXYWHBBoxCoder(**bc)
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.shared_fcs = nn.ModuleList([nn.Linear(in_channels * roi_feat_size ** 2, fc_out_channels),
                                         nn.Linear(fc_out_channels, fc_out_channels)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes + 1)
        self.fc_reg = nn.Linear(fc_out_channels, 4 if reg_class_agnostic else 4 * num_classes)
        for m in self.shared_fcs:
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
        nn.init.normal_(self.fc_cls.weight, std=0.01)
        nn.init.zeros_(self.fc_cls.bias)
        nn.init.normal_(self.fc_reg.weight, std=0.001)
        nn.init.zeros_(self.fc_reg.bias)

    def forward(self, x):
        x = x.flatten(1)
        # (vendor GEMMs: only ever inside _lib.blas_turn — one Stream-K kernel at a time)
        for fc in self.shared_fcs:
            x = _lib.blas_linear(x, fc.weight, fc.bias, relu=True)
        return _lib.blas_linear(x, self.fc_cls.weight, self.fc_cls.bias), \
            _lib.blas_linear(x, self.fc_reg.weight, self.fc_reg.bias)

    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, fused=None):
        """mmdet BBoxHead.loss: cls avg_factor = #(label_weights > 0); box loss over positives of
        the class-specific prediction, averaged over ALL sampled RoIs (bbox_targets.size(0))."""
        from ..mm3d.losses import FocalLoss, L1Loss
        if fused_on(fused) and cls_score.is_cuda and isinstance(self.loss_cls, FocalLoss) and self.loss_cls.gamma == 2.0 \
                and self.loss_cls.reduction == 'mean' and isinstance(self.loss_bbox, L1Loss) \
                and self.loss_bbox.reduction == 'mean':
            meta = (self.num_classes, bool(self.reg_class_agnostic), self.loss_cls.alpha,
                    self.loss_cls.loss_weight, self.loss_bbox.loss_weight)
            parts = _FusedBBoxHeadLoss.apply(cls_score.contiguous(), bbox_pred.contiguous(),
                                             labels.long().contiguous(), label_weights.float().contiguous(),
                                             bbox_targets.float().contiguous(),
                                             bbox_weights.float().contiguous(), meta)
            return dict(loss_cls=parts[0], acc=parts[2].detach(), loss_bbox=parts[1])
        n_valid = (label_weights > 0).sum().clamp(min=1).float()
        losses = dict()
        losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=n_valid)
        correct = (cls_score.argmax(dim=1) == labels).float() * (label_weights > 0).float()
        losses['acc'] = correct.sum() * 100.0 / n_valid
        pos = ((labels >= 0) & (labels < self.num_classes)).float() * (label_weights > 0).float()
        if self.reg_class_agnostic:
            pred = bbox_pred
        else:
            sel = labels.clamp(max=self.num_classes - 1)[:, None, None].expand(-1, 1, 4)
            pred = bbox_pred.view(bbox_pred.size(0), -1, 4).gather(1, sel).squeeze(1)
        losses['loss_bbox'] = self.loss_bbox(pred, bbox_targets, bbox_weights * pos[:, None],
                                             avg_factor=n_valid)
        return losses


class StandardRoIHead(nn.Module):

    def __init__(self, bbox_roi_extractor, bbox_head, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        rl = bbox_roi_extractor['roi_layer']
        self.out_size = rl.get('output_size', 7)
        self.sampling_ratio = rl.get('sampling_ratio', 0)
        self.aligned = rl.get('aligned', True)
        self.featmap_strides = list(bbox_roi_extractor['featmap_strides'])
        self.finest_scale = bbox_roi_extractor.get('finest_scale', 56)
        bh = dict(bbox_head)
        assert bh.pop('type') == 'Shared2FCBBoxHead'
        self.bbox_head = Shared2FCBBoxHead(**bh)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def extract(self, feats, rois):
        return roi_align_fpn(feats, rois, self.featmap_strides, self.out_size, self.sampling_ratio,
                             self.aligned, self.finest_scale)

    def _targets_device(self, proposals, gt_bboxes, gt_labels, keys):
        """dm_roi2d_targets: assignment, sampling and BBoxHead.get_targets of the batch in 4 launches."""
        cfg = self.train_cfg
        a_cfg, s_cfg = cfg['assigner'], cfg['sampler']
        num, nc = int(s_cfg['num']), self.bbox_head.num_classes
        props = [p.detach().float().contiguous() for p, _ in proposals]
        oks = [ok.contiguous() for _, ok in proposals]
        assert all(o.dtype == torch.bool for o in oks) and len({p.shape for p in props}) == 1
        gts, gt_ptrs, n_gt = _gt_arrays(gt_bboxes)
        labs = [l.detach().long().contiguous() for l in gt_labels]
        b, dev = len(props), props[0].device
        _lib.require_device(keys, *props)
        rois = torch.empty((b * num, 5), dtype=torch.float32, device=dev)
        labels = torch.empty(b * num, dtype=torch.int64, device=dev)
        lw = torch.empty(b * num, dtype=torch.float32, device=dev)
        tgts = torch.empty((b * num, 4), dtype=torch.float32, device=dev)
        bw = torch.empty((b * num, 4), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = _lib.workspace(L.dm_det2d_assign_workspace_bytes(b, int(keys.shape[1])), dev, 'det2d')
        coder = self.bbox_head.bbox_coder
        _lib.check(L.dm_roi2d_targets(
            _ptr_array(props), _ptr_array(oks), int(props[0].shape[0]), int(props[0].shape[1]), gt_ptrs,
            _ptr_array(labs), n_gt, b, int(bool(s_cfg.get('add_gt_as_proposals', True))), _lib.ptr(keys),
            int(keys.shape[1]), float(a_cfg['pos_iou_thr']), float(a_cfg['neg_iou_thr']),
            float(a_cfg['min_pos_iou']), int(bool(a_cfg.get('match_low_quality', False))), num,
            int(num * s_cfg['pos_fraction']), nc, _lib.floats(coder.means), _lib.floats(coder.stds),
            _lib.ptr(rois), _lib.ptr(labels), _lib.ptr(lw), _lib.ptr(tgts), _lib.ptr(bw), _lib.ptr(ws),
            ws.numel(), _lib.stream()), 'dm_roi2d_targets')
        return rois, labels, lw, tgts, bw

    def _targets_tensor(self, proposals, gt_bboxes, gt_labels, keys):
        """The same as dense tensor operations per image (comparator of the kernel; CPU host-logic tests)."""
        cfg = self.train_cfg
        a_cfg, s_cfg = cfg['assigner'], cfg['sampler']
        num = s_cfg['num']
        nc = self.bbox_head.num_classes
        rois, labels, lw, tgts, bw = [], [], [], [], []
        with torch.no_grad():
            for i, ((props, ok), gt, gl) in enumerate(zip(proposals, gt_bboxes, gt_labels)):
                boxes, valid = props[:, :4], ok
                if s_cfg.get('add_gt_as_proposals', True) and gt.shape[0] > 0:
                    boxes = torch.cat([gt.float(), boxes], dim=0)
                    valid = torch.cat([torch.ones_like(gt[:, 0], dtype=torch.bool), valid])
                assigned = max_iou_assign(boxes, gt.float(), a_cfg['pos_iou_thr'], a_cfg['neg_iou_thr'],
                                          a_cfg['min_pos_iou'], a_cfg.get('match_low_quality', False),
                                          box_valid=valid)
                ip, pos_ok, ineg, neg_ok = random_sample(assigned, num, s_cfg['pos_fraction'], keys=keys[i])
                idx = torch.cat([ip, ineg])
                okk = torch.cat([pos_ok, neg_ok])
                is_pos = torch.cat([pos_ok, torch.zeros_like(neg_ok)])
                # compact to `num` rows, sampled ones first (positives before negatives)
                front = (_lib.sort_rows((~okk).float(), descending=False) if okk.is_cuda and okk.numel() else
                         torch.sort((~okk).long(), stable=True)[1])[:num]
                idx, okk, is_pos = idx[front], okk[front], is_pos[front]
                b = boxes[idx]
                if gt.shape[0] > 0:
                    g = (assigned[idx] - 1).clamp(min=0)
                    t = self.bbox_head.bbox_coder.encode(b, gt.float()[g])
                    lab = torch.where(is_pos, gl[g], torch.full_like(gl[g], nc))
                else:
                    t = b.new_zeros((b.shape[0], 4))
                    lab = torch.full((b.shape[0],), nc, dtype=torch.long, device=b.device)
                t = torch.where(is_pos[:, None], t, torch.zeros_like(t))
                rois.append(torch.cat([b.new_full((b.shape[0], 1), float(i)), b], dim=1)
                            * okk[:, None].float())
                labels.append(lab), lw.append(okk.float()), tgts.append(t)
                bw.append(is_pos.float()[:, None].expand(-1, 4))
            rois = torch.cat(rois)
        return rois, torch.cat(labels), torch.cat(lw), torch.cat(tgts), torch.cat(bw)

    def forward_train(self, feats, img_metas, proposals, gt_bboxes, gt_labels, fused=None, keys=None):
        cfg = self.train_cfg
        a_cfg, s_cfg = cfg['assigner'], cfg['sampler']
        num = s_cfg['num']
        nc = self.bbox_head.num_classes
        if keys is None:
            keys = torch.rand((len(proposals), proposals[0][0].shape[0] + max(g.shape[0] for g in gt_bboxes)),
                              device=proposals[0][0].device)
        if fused_on(fused) and keys.is_cuda and len(proposals) <= 8 and max(g.shape[0] for g in gt_bboxes) <= 256 \
                and num <= 512:
            rois, labels, lw, tgts, bw = self._targets_device(proposals, gt_bboxes, gt_labels, keys)
            cls_score, bbox_pred = self.bbox_head(self.extract(feats, rois))
            return self.bbox_head.loss(cls_score, bbox_pred, labels, lw, tgts, bw)
        rois, labels, lw, tgts, bw = self._targets_tensor(proposals, gt_bboxes, gt_labels, keys)
        feats_roi = self.extract(feats, rois)
        cls_score, bbox_pred = self.bbox_head(feats_roi)
        return self.bbox_head.loss(cls_score, bbox_pred, labels, lw, tgts, bw, fused=fused)

    def simple_test_pre_nms(self, feats, proposals, img_metas):
        """test_mixins.simple_test_bboxes up to (not including) NMS: per image
        (decoded boxes (P, 4*C), scores (P, C+1)); rows of invalid proposals carry zero scores."""
        rois = torch.cat([torch.cat([p.new_full((p.shape[0], 1), float(i)), p[:, :4]], dim=1)
                          for i, (p, _) in enumerate(proposals)])
        cls_score, bbox_pred = self.bbox_head(self.extract(feats, rois))
        use_sigmoid = getattr(self.bbox_head.loss_cls, 'use_sigmoid', False)
        score = torch.sigmoid(cls_score) if use_sigmoid else F.softmax(cls_score, dim=-1)
        out, off = [], 0
        for i, (p, ok) in enumerate(proposals):
            n = p.shape[0]
            boxes = self.bbox_head.bbox_coder.decode(rois[off:off + n, 1:], bbox_pred[off:off + n],
                                                     max_shape=img_metas[i]['img_shape'])
            out.append((boxes, score[off:off + n] * ok[:, None].float()))
            off += n
        return out


# ------------------------------------------------------------------ detector
@DETECTORS.register_module()
class FasterRCNN(DetectorStepMixin, nn.Module):

    def __init__(self, backbone, neck, rpn_head, roi_head, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        bb = dict(backbone)
        assert bb.pop('type') == 'ResNet'
        self.backbone = ResNet(**bb)
        nk = dict(neck)
        assert nk.pop('type') == 'FPN'
        self.neck = FPN(**nk)
        rp = dict(rpn_head)
        assert rp.pop('type') == 'RPNHead'
        self.rpn_head = RPNHead(train_cfg=train_cfg['rpn'] if train_cfg else None,
                                test_cfg=test_cfg['rpn'] if test_cfg else None, **rp)
        rh = dict(roi_head)
        assert rh.pop('type') == 'StandardRoIHead'
        self.roi_head = StandardRoIHead(train_cfg=train_cfg['rcnn'] if train_cfg else None,
                                        test_cfg=test_cfg['rcnn'] if test_cfg else None, **rh)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def extract_feat(self, img):
        return self.neck(self.backbone(img))

    # ---- backbone + FPN + RPN convolutions as ONE C-ABI call per pass (chain.py / dense_chain.py) ------------------
    def _build_trunk_chain(self, img, train):
        """The op table of ResNet (frozen BatchNorm folded into the weight packs, shortcut sum + ReLU in the last
        GEMM of a bottleneck), FPN (top-down sum in the lateral GEMM's epilogue) and the RPN head (3x3 + ReLU, then
        objectness | deltas as one 1x1 GEMM per level), and — train — its mirror: mmdet ResNet / FPN / RPNHead
        forward as configured at configs/detmatch/001/detmatch/split_0.py:39-99.  Outputs: the five pyramid maps,
        then the five raw head maps (what `extract_feat` + `rpn_head` hand to prefetch_trunk / the inference path)."""
        from ..dense_chain import DenseChainBuilder, _watch_loads
        bb, neck, rpn = self.backbone, self.neck, self.rpn_head
        b = DenseChainBuilder('faster_rcnn.trunk.%s' % ('train' if train else 'eval'), img.device, training=train)
        n, _, h, w = img.shape
        x = b.input(n, 4, h, w, False)

        def cbn(t, conv, bn, relu, residual=None):
            # the layer's own cached (scale, shift) tensors (FrozenBN.scale_shift: torch.rsqrt differs from the device
            # rsqrtf in the last bit, and the chain must equal the op-by-op path bit for bit); they are constants
            # until FrozenBN.GENERATION / a load_state_dict moves them: the chain keeps copies at stable addresses and
            # re-copies them when either counter moves
            s_ = b.weights.mirror(lambda bn=bn: bn.scale_shift()[0])
            sh_ = b.weights.mirror(lambda bn=bn: bn.scale_shift()[1])
            return b.conv(t, conv.weight, sh_, tuple(conv.stride), tuple(conv.padding), relu=relu, w_scale=s_,
                          residual=residual, bias_trainable=False)

        x = cbn(x, bb.conv1, bb.bn1, True)
        x = b.maxpool(x, 3, 2, 1)
        feats = []
        for i, name in enumerate(bb.res_layers):
            for blk in getattr(bb, name):
                out = cbn(x, blk.conv1, blk.bn1, True)
                out = cbn(out, blk.conv2, blk.bn2, True)
                idt = x if blk.downsample is None else cbn(x, blk.downsample[0], blk.downsample[1], False)
                x = cbn(out, blk.conv3, blk.bn3, True, residual=idt)
            if i in bb.out_indices:
                feats.append(x)
        assert len(feats) == neck.num_ins

        def plain(t, m, relu=False, residual=None):
            return b.conv(t, m.weight, m.bias, tuple(m.stride), tuple(m.padding), relu=relu, residual=residual)

        lat = [None] * len(feats)
        lat[-1] = plain(feats[-1], neck.lateral_convs[-1].conv)
        for i in range(len(feats) - 2, -1, -1):
            up = b.resize_nearest(lat[i + 1], feats[i].h, feats[i].w)
            lat[i] = plain(feats[i], neck.lateral_convs[i].conv, residual=up)
        outs = [plain(t, c.conv) for c, t in zip(neck.fpn_convs, lat)]
        for _ in range(neck.num_outs - len(outs)):
            outs.append(b.maxpool(outs[-1], 1, 2, 0))
        a = rpn.num_anchors
        total = 5 * a + (-5 * a) % 4
        wcat = b.weights.cat([rpn.rpn_cls.weight, rpn.rpn_reg.weight], pad_to=total)
        bcat = b.weights.cat([rpn.rpn_cls.bias, rpn.rpn_reg.bias], pad_to=total)
        wparts = [(rpn.rpn_cls.weight, 0, a), (rpn.rpn_reg.weight, a, 5 * a)]
        bparts = [(rpn.rpn_cls.bias, 0, a), (rpn.rpn_reg.bias, a, 5 * a)]
        raws = []
        for t in outs:
            t2 = plain(t, rpn.rpn_conv, relu=True)
            raws.append(b.conv(t2, wcat, bcat, (1, 1), (0, 0), weight_parts=wparts, bias_parts=bparts))
        for t in outs + raws:
            b.output(t)
        _watch_loads([self])
        from .backbone import FrozenBN
        b.weights.mirror_key = lambda: (FrozenBN.GENERATION, dense_conv.LOAD_EPOCH[0])
        return b.build()

    def _trunk_chain(self, img):
        """-> (pyramid maps, raw RPN maps) through the chain, or None when it does not apply."""
        from .. import chain as _chain
        from .backbone import FrozenBN
        if not _chain.on('trunk2d') or not img.is_cuda or img.dtype != torch.float32 or img.shape[1] != 3:
            return None
        train = torch.is_grad_enabled() and not self._frozen()
        if not _chain.on('trunk2d', train):
            return None
        if train and not self.training:
            return None
        if self.neck.num_outs < self.neck.num_ins or self.backbone.conv1.weight.requires_grad:
            return None
        key = (bool(train), tuple(img.shape), dense_conv.get_math_code())
        cache = self.__dict__.setdefault('_trunk_chains', {})
        ch = cache.get(key)
        if ch is None or not ch.valid():
            if len(cache) > 8:
                cache.clear()
            ch = cache[key] = self._build_trunk_chain(img, train)
        x4 = dense_conv._pad_channels(dense_conv._cl(img.detach()), 4)
        outs = ch(x4)
        k = len(outs) // 2
        return tuple(outs[:k]), list(outs[k:])

    def _trunk(self, img):
        """extract_feat + rpn_head -> (x, cls, reg, raw)."""
        hit = self._trunk_chain(img)
        a = self.rpn_head.num_anchors
        if hit is not None:
            x, raw = hit
            self.rpn_head._raw_levels = raw
            return x, [y[:, :a] for y in raw], [y[:, a:5 * a] for y in raw], raw
        x = self.extract_feat(img)
        cls, reg = self.rpn_head(x)
        return x, cls, reg, self.rpn_head._raw_levels

    # ---- one trunk pass for several forward_train calls of an iteration (scheduling only) --------------------
    # The backbone's BatchNorm is frozen (eval), FPN and RPN have none: the samples of a batch are independent, so
    # backbone + FPN + RPN convolutions of the labeled and the unlabeled images of a DetMatch iteration can run as
    # ONE batch (half the launches, fuller tiles on the small pyramid levels) although their losses are formed at
    # different times.  The trunk outputs are cut from the graph: every forward_train call gets its OWN leaves, the
    # batch slices of the detached outputs that belong to its images (views, no copies), and back-propagates into
    # their .grad whenever its backward runs; finish_deferred_backward() then assembles the full-batch gradients
    # (one copy per span into one buffer per output) and sends them through the trunk once.  (Leaves of the whole
    # batch, sliced per call, made autograd build every span's gradient as zeros(full) + copy and add the spans up:
    # 82 launches and 0.55 ms of full-size fills, strided copies and adds per iteration.)
    def prefetch_trunk(self, imgs):
        """imgs: image batches (B_i, 3, H, W) of equal H, W that forward_train will be called with (same tensor
        objects) before finish_deferred_backward()."""
        self._shared = None
        if len(imgs) < 2 or len({tuple(i.shape[1:]) for i in imgs}) != 1 or not torch.is_grad_enabled() \
                or not self.training:
            return False
        batch = torch.cat([i for i in imgs], dim=0)
        x, _, _, raw = self._trunk(batch)
        outs = list(x) + list(raw)
        cut = [t.detach() for t in outs]
        spans, lo = {}, 0
        for i in imgs:
            hi = lo + i.shape[0]
            spans[id(i)] = (i, lo, hi, [t[lo:hi].requires_grad_(True) for t in cut])
            lo = hi
        self._shared = dict(outs=outs, spans=spans, n_feat=len(x))
        return True

    def _shared_slices(self, img):
        sh = getattr(self, '_shared', None)
        hit = sh['spans'].get(id(img)) if sh else None
        if hit is None or hit[0] is not img:
            return None
        leaves = hit[3]
        nf = sh['n_feat']
        x = tuple(leaves[:nf])
        raw = list(leaves[nf:])
        a = self.rpn_head.num_anchors
        self.rpn_head._raw_levels = raw
        return x, [y[:, :a] for y in raw], [y[:, a:5 * a] for y in raw]

    def finish_deferred_backward(self):
        """Trunk backward of a prefetch_trunk() pass with everything the heads have accumulated so far."""
        sh = getattr(self, '_shared', None)
        self._shared = None
        if not sh:
            return
        pairs = []
        spans = sorted(sh['spans'].values(), key=lambda v: v[1])
        for k, o in enumerate(sh['outs']):
            grads = [(lo, hi, leaves[k].grad) for _, lo, hi, leaves in spans]
            if not o.requires_grad or all(g is None for _, _, g in grads):
                continue
            full = torch.empty_like(o)      # (the outputs' own strides: NHWC memory for the feature maps)
            for lo, hi, g in grads:
                if g is None:
                    full[lo:hi].zero_()
                else:
                    full[lo:hi].copy_(g)
            pairs.append((o, full))
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, **kwargs):
        """mmdet TwoStageDetector.forward_train -> loss_rpn_cls, loss_rpn_bbox, loss_cls, acc, loss_bbox."""
        assert gt_bboxes_ignore is None
        shared = self._shared_slices(img)
        if shared is not None:
            x, cls, reg = shared
        else:
            x, cls, reg, _ = self._trunk(img)
        losses = self.rpn_head.loss(cls, reg, gt_bboxes, img_metas)
        proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg['rpn'] if self.test_cfg else None)
        proposals = self.rpn_head.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg],
                                             img_metas, proposal_cfg)
        losses.update(self.roi_head.forward_train(x, img_metas, proposals, gt_bboxes, gt_labels))
        return losses

    def _frozen(self):
        """No parameter of the trunk requires grad (the EMA teacher): its forward records no autograd graph."""
        sentinel = self.rpn_head.rpn_conv.weight
        hit = self.__dict__.get('_frozen_state')
        if hit is None or hit[0] != sentinel.requires_grad:
            trunk = list(self.backbone.parameters()) + list(self.neck.parameters()) + list(self.rpn_head.parameters())
            hit = self.__dict__['_frozen_state'] = (sentinel.requires_grad, not any(p.requires_grad for p in trunk))
        return hit[1]

    def simple_test_pre_nms(self, img, img_metas):
        """The body of SimpleTest_2D.forward (processors_2d.py:36-84)."""
        x, cls, reg, _ = self._trunk(img)
        proposals = self.rpn_head.get_bboxes(cls, reg, img_metas, self.test_cfg['rpn'])
        return self.roi_head.simple_test_pre_nms(x, proposals, img_metas)

    @torch.no_grad()
    def simple_test(self, img, img_metas, proposals=None, rescale=False, **kwargs):
        """mmdet TwoStageDetector.simple_test -> per image, per class (k, 5) numpy arrays."""
        from ..mm3d.bbox_utils import modified_multiclass_nms
        rc = self.test_cfg['rcnn']
        nc = self.roi_head.bbox_head.num_classes
        results = []
        for (boxes, scores), meta in zip(self.simple_test_pre_nms(img, img_metas), img_metas):
            if rescale:
                sf = boxes.new_tensor(meta['scale_factor'])
                boxes = (boxes.view(boxes.size(0), -1, 4) / sf).view(boxes.size(0), -1)
            dets, labels, _ = modified_multiclass_nms(boxes, scores, rc['score_thr'], rc['nms'],
                                                      rc['max_per_img'])
            results.append([dets[labels == c].cpu().numpy() for c in range(nc)])
        return results
