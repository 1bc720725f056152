"""AnchorHeadSingle (+ anchor generation, axis-aligned target assignment) and
PointHeadSimple.

Reference: pcdet/models/dense_heads/{anchor_head_template,anchor_head_single,
point_head_template,point_head_simple}.py and
target_assigner/{anchor_generator,axis_aligned_target_assigner}.py.

Same arithmetic as the reference, restated WITHOUT its host round-trips
(`.cpu().numpy().argmax` :144,:149, `cur_gt[k].sum() == 0` loops :55, per-class boolean
indexing, `tb_dict[...] = x.item()`): target assignment is a dense masked formulation over
the padded GT tensor, so a training step issues no device->host copy here.  `tb_dict`
values are detached 0-d tensors (call .item() on them outside the step if wanted).
"""

import numpy as np
import torch
import torch.nn as nn

from .. import dense_conv

from .. import roiaware_pool3d
from ..devconst import const
from . import utils as U
from ..bn_relu import fc_rows
from ..fused import on as fused_on


# ------------------------------------------------------------------- anchors
class AnchorGenerator(object):
    """anchor_generator.py:4-60; anchors per class: (1, ny, nx, n_size, n_rot, 7)."""

    def __init__(self, anchor_range, anchor_generator_config):
        self.anchor_range = anchor_range
        self.anchor_sizes = [c['anchor_sizes'] for c in anchor_generator_config]
        self.anchor_rotations = [c['anchor_rotations'] for c in anchor_generator_config]
        self.anchor_heights = [c['anchor_bottom_heights'] for c in anchor_generator_config]
        self.align_center = [c.get('align_center', False) for c in anchor_generator_config]
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes):
        assert len(grid_sizes) == self.num_of_anchor_sets
        all_anchors, num_per_loc = [], []
        r = self.anchor_range
        for grid_size, anchor_size, anchor_rotation, anchor_height, align_center in zip(
                grid_sizes, self.anchor_sizes, self.anchor_rotations, self.anchor_heights,
                self.align_center):
            num_per_loc.append(len(anchor_rotation) * len(anchor_size) * len(anchor_height))
            if align_center:
                x_stride = (r[3] - r[0]) / grid_size[0]
                y_stride = (r[4] - r[1]) / grid_size[1]
                x_offset, y_offset = x_stride / 2, y_stride / 2
            else:
                x_stride = (r[3] - r[0]) / (grid_size[0] - 1)
                y_stride = (r[4] - r[1]) / (grid_size[1] - 1)
                x_offset, y_offset = 0, 0
            x_shifts = torch.arange(r[0] + x_offset, r[3] + 1e-5, step=x_stride, dtype=torch.float32)
            y_shifts = torch.arange(r[1] + y_offset, r[4] + 1e-5, step=y_stride, dtype=torch.float32)
            z_shifts = x_shifts.new_tensor(anchor_height)
            n_size, n_rot = len(anchor_size), len(anchor_rotation)
            rot = x_shifts.new_tensor(anchor_rotation)
            size = x_shifts.new_tensor(anchor_size)
            xs, ys, zs = torch.meshgrid([x_shifts, y_shifts, z_shifts], indexing='ij')
            anchors = torch.stack((xs, ys, zs), dim=-1)
            anchors = anchors[:, :, :, None, :].repeat(1, 1, 1, n_size, 1)
            size = size.view(1, 1, 1, -1, 3).repeat([*anchors.shape[0:3], 1, 1])
            anchors = torch.cat((anchors, size), dim=-1)
            anchors = anchors[:, :, :, :, None, :].repeat(1, 1, 1, 1, n_rot, 1)
            rot = rot.view(1, 1, 1, 1, -1, 1).repeat([*anchors.shape[0:3], n_size, 1, 1])
            anchors = torch.cat((anchors, rot), dim=-1)
            anchors = anchors.permute(2, 1, 0, 3, 4, 5).contiguous()
            anchors[..., 2] += anchors[..., 5] / 2  # bottom height -> box centre
            all_anchors.append(anchors)
        return all_anchors, num_per_loc


def valid_gt_mask(gt_boxes):
    """(B, M, 7+) -> (B, M) bool: rows up to the last non-zero row, at least the first row
    (the reference's trailing-zero trim, axis_aligned_target_assigner.py:53-58)."""
    nz = gt_boxes.sum(dim=-1) != 0
    # position of the last non-zero row (reference tests `cur_gt[cnt].sum() == 0`)
    idx = torch.arange(gt_boxes.shape[1], device=gt_boxes.device)[None, :]
    last = torch.where(nz, idx, torch.zeros_like(idx)).max(dim=1, keepdim=True)[0]
    return idx <= last


class AxisAlignedTargetAssigner(object):
    """axis_aligned_target_assigner.py:8-209 (POS_FRACTION < 0, MATCH_HEIGHT False,
    NORM_BY_NUM_EXAMPLES False: the only settings configs/detmatch use)."""

    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        anchor_generator_cfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        anchor_target_cfg = model_cfg.TARGET_ASSIGNER_CONFIG
        assert anchor_target_cfg.POS_FRACTION < 0 and not match_height
        assert not anchor_target_cfg.NORM_BY_NUM_EXAMPLES
        self.box_coder = box_coder
        self.class_names = list(class_names)
        self.anchor_class_names = [c['class_name'] for c in anchor_generator_cfg]
        self.matched_thresholds = {c['class_name']: c['matched_threshold']
                                   for c in anchor_generator_cfg}
        self.unmatched_thresholds = {c['class_name']: c['unmatched_threshold']
                                     for c in anchor_generator_cfg}

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: [(1,ny,nx,1,n_rot,7)] per class; gt (B, M, 8) -> dict of
        box_cls_labels (B, A) int32, box_reg_targets (B, A, 7), reg_weights (B, A).

        All anchor classes and all samples in ONE set of tensor ops ((C, B, A, M) overlaps):
        `assign_targets_single` below is the same computation for one (class, sample) and is what
        the tests compare this against; the reference loops samples x classes in Python with
        .cpu().numpy() argmax read-backs (axis_aligned_target_assigner.py:36-130)."""
        shapes = {tuple(a.shape) for a in all_anchors}
        if len(shapes) != 1:         # per-class feature maps differ: fall back to the loop
            return self._assign_targets_loop(all_anchors, gt_boxes_with_classes)
        if fused_on() and all_anchors[0].is_cuda and gt_boxes_with_classes.shape[-1] == 8:
            return self._assign_targets_device(all_anchors, gt_boxes_with_classes)
        gt_classes = gt_boxes_with_classes[:, :, -1].int()
        gt_boxes = gt_boxes_with_classes[:, :, :-1]
        B, M = gt_boxes.shape[:2]
        C = len(all_anchors)
        fmap = all_anchors[0].shape[:3]
        anchors = torch.stack([a.view(-1, a.shape[-1]) for a in all_anchors])       # (C, A, 7)
        A = anchors.shape[1]
        valid = valid_gt_mask(gt_boxes)
        num_class = len(self.class_names)
        cls_eff = torch.where(gt_classes == 0, torch.full_like(gt_classes, num_class), gt_classes)
        dev, dt = anchors.device, anchors.dtype
        cids = const([self.class_names.index(n) + 1 for n in self.anchor_class_names], dev, torch.int32)
        matched = const([self.matched_thresholds[n] for n in self.anchor_class_names], dev, dt)
        unmatched = const([self.unmatched_thresholds[n] for n in self.anchor_class_names], dev, dt)
        sel = valid[None] & (cls_eff[None] == cids[:, None, None])                  # (C, B, M)
        # nearest-BEV IoU (box_utils.py:286-298), broadcast to (C, B, A, M)
        ab = U.boxes3d_lidar_to_aligned_bev_boxes(anchors.view(-1, anchors.shape[-1])[:, 0:7]).view(C, 1, A, 1, 4)
        gb = U.boxes3d_lidar_to_aligned_bev_boxes(gt_boxes.reshape(-1, gt_boxes.shape[-1])[:, 0:7]).view(1, B, 1, M, 4)
        x_len = torch.clamp_min(torch.min(ab[..., 2], gb[..., 2]) - torch.max(ab[..., 0], gb[..., 0]), min=0)
        y_len = torch.clamp_min(torch.min(ab[..., 3], gb[..., 3]) - torch.max(ab[..., 1], gb[..., 1]), min=0)
        area_a = (ab[..., 2] - ab[..., 0]) * (ab[..., 3] - ab[..., 1])
        area_b = (gb[..., 2] - gb[..., 0]) * (gb[..., 3] - gb[..., 1])
        inter = x_len * y_len
        overlap = inter / torch.clamp_min(area_a + area_b - inter, min=1e-6)        # (C, B, A, M)
        overlap = torch.where(sel[:, :, None, :], overlap, overlap.new_full((), -1.0))
        a2g_max, a2g_arg = overlap.max(dim=3)                                        # (C, B, A)
        g2a_max = overlap.max(dim=2)[0]                                              # (C, B, M)
        g2a_max = torch.where((g2a_max == 0) | ~sel, g2a_max.new_full((), -2.0), g2a_max)
        force = (overlap == g2a_max[:, :, None, :]).any(dim=3)                       # :154
        cls_of_anchor = torch.gather(gt_classes[None].expand(C, B, M), 2, a2g_arg)
        labels = torch.full((C, B, A), -1, dtype=torch.int32, device=dev)
        labels = torch.where(force, cls_of_anchor, labels)                           # :156
        labels = torch.where(a2g_max >= matched[:, None, None], cls_of_anchor, labels)   # :161
        labels = torch.where(a2g_max < unmatched[:, None, None], torch.zeros_like(labels), labels)
        labels = torch.where(force, cls_of_anchor, labels)                           # :188
        labels = torch.where(sel.any(dim=2)[:, :, None], labels, torch.zeros_like(labels))
        fg = labels > 0
        idx7 = a2g_arg[..., None].expand(C, B, A, 7)
        matched_gt = torch.gather(gt_boxes[None, :, :, 0:7].expand(C, B, M, 7), 2, idx7)
        enc = self.box_coder.encode_torch(matched_gt.reshape(-1, 7),
                                          anchors[:, None, :, 0:7].expand(C, B, A, 7).reshape(-1, 7))
        enc = enc.view(C, B, A, -1)
        tgt = torch.where(fg[..., None], enc, torch.zeros_like(enc))
        code = self.box_coder.code_size
        # per location the classes are concatenated: (B, ny, nx, [z], C * n_rot)
        per_loc = A // (fmap[0] * fmap[1] * fmap[2])
        to_loc = lambda t, tail: t.view(C, B, *fmap, per_loc, *tail).permute(
            1, 2, 3, 4, 0, 5, *range(6, 6 + len(tail))).reshape(B, -1, *tail)
        return {'box_cls_labels': to_loc(labels, ()), 'box_reg_targets': to_loc(tgt, (code,)),
                'reg_weights': to_loc(fg.to(dt), ())}

    def _assign_targets_device(self, all_anchors, gt):
        """dm_anchor_assign (csrc/anchor_assign.hip): two launches for the whole batch.  The BEV
        rectangles come from the same tensor ops as the reference (anchors: once, cached)."""
        from .. import _lib
        dev = all_anchors[0].device
        key = tuple(a.data_ptr() for a in all_anchors)
        cache = getattr(self, '_dev_cache', None)
        if cache is None or cache[0] != key:
            anchors = torch.stack([a.reshape(-1, a.shape[-1])[:, 0:7] for a in all_anchors]).contiguous()
            abev = U.boxes3d_lidar_to_aligned_bev_boxes(anchors.view(-1, 7)).view(len(all_anchors), -1, 4).contiguous()
            cids = const([self.class_names.index(n) + 1 for n in self.anchor_class_names], dev, torch.int32)
            matched = const([self.matched_thresholds[n] for n in self.anchor_class_names], dev, torch.float32)
            unmatched = const([self.unmatched_thresholds[n] for n in self.anchor_class_names], dev, torch.float32)
            fmap = all_anchors[0].shape[:3]
            cache = self._dev_cache = (key, anchors, abev, cids, matched, unmatched,
                                       anchors.shape[1] // (fmap[0] * fmap[1] * fmap[2]))
        _, anchors, abev, cids, matched, unmatched, per_loc = cache
        gt = gt.detach().float().contiguous()
        B, M = gt.shape[:2]
        C, A = anchors.shape[:2]
        gbev = U.boxes3d_lidar_to_aligned_bev_boxes(gt.view(-1, 8)[:, 0:7]).contiguous() if M else gt.new_zeros((0, 4))
        labels = torch.empty((B, A * C), dtype=torch.int32, device=dev)
        targets = torch.empty((B, A * C, self.box_coder.code_size), dtype=torch.float32, device=dev)
        weights = torch.empty((B, A * C), dtype=torch.float32, device=dev)
        assert self.box_coder.code_size == 7
        L = _lib.lib()
        ws = _lib.workspace(L.dm_anchor_assign_workspace_bytes(B, M, C), dev, 'anchor_assign')
        _lib.check(L.dm_anchor_assign(_lib.ptr(anchors), _lib.ptr(abev), _lib.ptr(gt), _lib.ptr(gbev),
                                      _lib.ptr(cids), _lib.ptr(matched), _lib.ptr(unmatched), B, M, C, A,
                                      per_loc, len(self.class_names), _lib.ptr(labels), _lib.ptr(targets),
                                      _lib.ptr(weights), _lib.ptr(ws), ws.numel(), _lib.stream()),
                   'dm_anchor_assign')
        return {'box_cls_labels': labels, 'box_reg_targets': targets, 'reg_weights': weights}

    def _assign_targets_loop(self, all_anchors, gt_boxes_with_classes):
        gt_classes = gt_boxes_with_classes[:, :, -1].int()
        gt_boxes = gt_boxes_with_classes[:, :, :-1]
        B = gt_boxes.shape[0]
        valid = valid_gt_mask(gt_boxes)
        num_class = len(self.class_names)
        # class id 0 (padding inside the trimmed range) indexes class_names[-1] in the reference
        cls_eff = torch.where(gt_classes == 0, torch.full_like(gt_classes, num_class), gt_classes)
        labels_all, targets_all, weights_all = [], [], []
        for anchor_class_name, anchors in zip(self.anchor_class_names, all_anchors):
            feature_map_size = anchors.shape[:3]
            flat = anchors.view(-1, anchors.shape[-1])
            cid = self.class_names.index(anchor_class_name) + 1
            sel = valid & (cls_eff == cid)
            lab, tgt, w = [], [], []
            for k in range(B):
                t = self.assign_targets_single(
                    flat, gt_boxes[k], gt_classes[k], sel[k],
                    self.matched_thresholds[anchor_class_name],
                    self.unmatched_thresholds[anchor_class_name])
                lab.append(t[0].view(*feature_map_size, -1))
                tgt.append(t[1].view(*feature_map_size, -1, self.box_coder.code_size))
                w.append(t[2].view(*feature_map_size, -1))
            labels_all.append(torch.stack(lab))
            targets_all.append(torch.stack(tgt))
            weights_all.append(torch.stack(w))
        code = self.box_coder.code_size
        return {'box_cls_labels': torch.cat(labels_all, dim=-1).view(B, -1),
                'box_reg_targets': torch.cat(targets_all, dim=-2).view(B, -1, code),
                'reg_weights': torch.cat(weights_all, dim=-1).view(B, -1)}

    def assign_targets_single(self, anchors, gt_boxes, gt_classes, sel, matched_threshold,
                              unmatched_threshold):
        """Dense form of :132-209.  `sel` (M) marks the GT rows of this anchor class."""
        num_anchors = anchors.shape[0]
        any_gt = sel.any()
        overlap = U.boxes3d_nearest_bev_iou(anchors[:, 0:7], gt_boxes[:, 0:7])       # (A, M)
        overlap = torch.where(sel[None, :], overlap, overlap.new_full((), -1.0))
        anchor_to_gt_max, anchor_to_gt_argmax = overlap.max(dim=1)
        gt_to_anchor_max = overlap.max(dim=0)[0]
        # :151-152 empty_gt_mask -> -1; unselected columns can never match either
        gt_to_anchor_max = torch.where((gt_to_anchor_max == 0) | ~sel,
                                       gt_to_anchor_max.new_full((), -2.0), gt_to_anchor_max)
        force = (overlap == gt_to_anchor_max[None, :]).any(dim=1)                    # :154
        cls_of_anchor = gt_classes[anchor_to_gt_argmax]
        labels = torch.full((num_anchors,), -1, dtype=torch.int32, device=anchors.device)
        labels = torch.where(force, cls_of_anchor, labels)                            # :156
        pos = anchor_to_gt_max >= matched_threshold
        labels = torch.where(pos, cls_of_anchor, labels)                              # :161
        bg = anchor_to_gt_max < unmatched_threshold
        labels = torch.where(bg, torch.zeros_like(labels), labels)                    # :187
        labels = torch.where(force, cls_of_anchor, labels)                            # :188
        labels = torch.where(any_gt, labels, torch.zeros_like(labels))                # :184-185
        fg = labels > 0
        enc = self.box_coder.encode_torch(gt_boxes[anchor_to_gt_argmax, 0:7], anchors[:, 0:7])
        bbox_targets = torch.where(fg[:, None], enc, torch.zeros_like(enc))
        return labels, bbox_targets, fg.to(anchors.dtype)


class _FusedAnchorHeadLoss(torch.autograd.Function):
    """dm_anchor_head_loss_forward / _backward (csrc/anchor_loss.hip): the classification, regression and
    direction losses of the anchor head in one launch each way -> tensor [cls, loc, dir]."""

    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, num_pos, meta):
        from .. import _lib
        L = _lib.lib()
        _lib.require_device(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, num_pos)
        b, a = int(labels.shape[0]), int(labels.shape[1])
        n_cls, n_bins, alpha, beta, dir_offset, w3, cw7 = meta
        out = torch.empty(3, dtype=torch.float32, device=cls_preds.device)
        ws = _lib.workspace(L.dm_anchor_head_loss_workspace_bytes(b, a), cls_preds.device, 'anchor_loss')
        _lib.check(L.dm_anchor_head_loss_forward(
            _lib.ptr(cls_preds), _lib.ptr(box_preds), _lib.ptr(dir_preds), _lib.ptr(labels), _lib.ptr(reg_targets),
            _lib.ptr(anchors), _lib.ptr(num_pos), b, a, n_cls, n_bins, alpha, beta, dir_offset, _lib.floats(w3),
            _lib.floats(cw7), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_anchor_head_loss_forward')
        ctx.save_for_backward(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, num_pos)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, grad):
        from .. import _lib
        L = _lib.lib()
        cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, num_pos = ctx.saved_tensors
        n_cls, n_bins, alpha, beta, dir_offset, w3, cw7 = ctx.meta
        b, a = int(labels.shape[0]), int(labels.shape[1])
        grad = grad.contiguous().float()
        g_cls, g_box = torch.empty_like(cls_preds), torch.empty_like(box_preds)
        g_dir = torch.empty_like(dir_preds) if dir_preds is not None else None
        _lib.check(L.dm_anchor_head_loss_backward(
            _lib.ptr(cls_preds), _lib.ptr(box_preds), _lib.ptr(dir_preds), _lib.ptr(labels), _lib.ptr(reg_targets),
            _lib.ptr(anchors), _lib.ptr(num_pos), b, a, n_cls, n_bins, alpha, beta, dir_offset, _lib.floats(w3),
            _lib.floats(cw7), _lib.ptr(grad), _lib.ptr(g_cls), _lib.ptr(g_box), _lib.ptr(g_dir), _lib.stream()),
            'dm_anchor_head_loss_backward')
        return g_cls, g_box, g_dir, None, None, None, None, None


class AnchorHeadSingle(nn.Module):
    """anchor_head_template.py:11-275 + anchor_head_single.py:7-75."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size,
                 point_cloud_range, predict_boxes_when_training=True, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        anchor_target_cfg = model_cfg.TARGET_ASSIGNER_CONFIG
        assert anchor_target_cfg.BOX_CODER == 'ResidualCoder'
        self.box_coder = U.ResidualCoder(**anchor_target_cfg.get('BOX_CODER_CONFIG', {}))
        gen_cfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        gen = AnchorGenerator(anchor_range=point_cloud_range, anchor_generator_config=gen_cfg)
        gs = np.array(grid_size)
        feature_map_size = [gs[:2] // c['feature_map_stride'] for c in gen_cfg]
        anchors, per_loc = gen.generate_anchors(feature_map_size)
        # buffers move with .to(device); the reference hard-codes .cuda() at construction (:31)
        self._n_anchor_sets = len(anchors)
        for i, a in enumerate(anchors):
            self.register_buffer('anchors_%d' % i, a, persistent=False)
        self.num_anchors_per_location = sum(per_loc)
        assert anchor_target_cfg.NAME == 'AxisAlignedTargetAssigner'
        self.target_assigner = AxisAlignedTargetAssigner(
            model_cfg=model_cfg, class_names=class_names, box_coder=self.box_coder,
            match_height=anchor_target_cfg.MATCH_HEIGHT)
        self.forward_ret_dict = {}
        lw = model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        self.cls_loss_func = U.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)
        self.reg_loss_func = U.WeightedSmoothL1Loss(code_weights=lw['code_weights'])
        self.dir_loss_func = U.WeightedCrossEntropyLoss()
        self.conv_cls = nn.Conv2d(input_channels, self.num_anchors_per_location * num_class, 1)
        self.conv_box = nn.Conv2d(input_channels,
                                  self.num_anchors_per_location * self.box_coder.code_size, 1)
        if model_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels,
                                          self.num_anchors_per_location * model_cfg.NUM_DIR_BINS, 1)
        else:
            self.conv_dir_cls = None
        pi = 0.01
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    @property
    def anchors(self):
        return [getattr(self, 'anchors_%d' % i) for i in range(self._n_anchor_sets)]

    def _cat_anchors(self):
        return torch.cat(self.anchors, dim=-3)

    def conv_heads(self, x):
        """The three 1x1 heads (anchor_head_single.py:20-37) as ONE implicit GEMM over the NHWC feature map:
        weights / biases concatenated along the output channels (padded to a multiple of 4 for the kernel's
        16-byte rows), the (B, H, W, 18 | 42 | 12) predictions are column slices of its output — the
        reference's permute(0, 2, 3, 1) is the layout we already have."""
        heads = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        widths = [m.out_channels for m in heads]
        pad = (-sum(widths)) % 4
        w = torch.cat([m.weight for m in heads], dim=0)
        b = torch.cat([m.bias for m in heads], dim=0)
        if pad:
            w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)
            b = torch.cat([b, b.new_zeros(pad)], dim=0)
        y = dense_conv.conv2d(x, w, b, 1, 0).permute(0, 2, 3, 1)          # (B, H, W, sum) NHWC view
        return [t.contiguous() for t in torch.split(y, widths + ([pad] if pad else []), dim=-1)[:len(heads)]]

    def forward(self, data_dict):
        outs = data_dict.pop('dense_head_convs', None)      # issued with the BEV backbone (backbones_2d._graphed)
        if outs is None:
            outs = self.conv_heads(data_dict['spatial_features_2d'])
        cls_preds, box_preds = outs[0], outs[1]
        self.forward_ret_dict['cls_preds'] = cls_preds
        self.forward_ret_dict['box_preds'] = box_preds
        dir_cls_preds = None
        if self.conv_dir_cls is not None:
            dir_cls_preds = outs[2]
            self.forward_ret_dict['dir_cls_preds'] = dir_cls_preds
        if self.training:
            self.forward_ret_dict.update(
                self.target_assigner.assign_targets(self.anchors, data_dict['gt_boxes']))
        if not self.training or self.predict_boxes_when_training:
            batch_cls_preds, batch_box_preds = self.generate_predicted_boxes(
                data_dict['batch_size'], cls_preds, box_preds, dir_cls_preds)
            data_dict['batch_cls_preds'] = batch_cls_preds
            data_dict['batch_box_preds'] = batch_box_preds
            data_dict['cls_preds_normalized'] = False
        return data_dict

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        """anchor_head_template.py:225-272"""
        anchors = self._cat_anchors()
        num_anchors = anchors.view(-1, anchors.shape[-1]).shape[0]
        batch_cls_preds = cls_preds.view(batch_size, num_anchors, -1).float()
        if fused_on() and box_preds.is_cuda and anchors.shape[-1] == 7 and self.box_coder.code_size == 7 \
                and box_preds.dtype == torch.float32 and getattr(self, 'boxes_detached_downstream', False):
            return batch_cls_preds, self._decode_device(batch_size, num_anchors, anchors, box_preds, dir_cls_preds)
        batch_anchors = anchors.view(1, -1, anchors.shape[-1]).repeat(batch_size, 1, 1)
        batch_box_preds = self.box_coder.decode_torch(box_preds.view(batch_size, num_anchors, -1),
                                                      batch_anchors)
        if dir_cls_preds is not None:
            dir_offset = self.model_cfg.DIR_OFFSET
            dir_limit_offset = self.model_cfg.DIR_LIMIT_OFFSET
            dir_labels = torch.max(dir_cls_preds.view(batch_size, num_anchors, -1), dim=-1)[1]
            period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS
            dir_rot = U.limit_period(batch_box_preds[..., 6] - dir_offset, dir_limit_offset, period)
            rot = dir_rot + dir_offset + period * dir_labels.to(batch_box_preds.dtype)
            batch_box_preds = torch.cat([batch_box_preds[..., :6], rot.unsqueeze(-1),
                                         batch_box_preds[..., 7:]], dim=-1)
        return batch_cls_preds, batch_box_preds

    @torch.no_grad()
    def _decode_device(self, batch_size, num_anchors, anchors, box_preds, dir_cls_preds):
        """The same in one launch (csrc/box_decode.hip, bit-identical); no autograd graph — only taken
        when the detector declares that nothing differentiates through the decoded boxes."""
        from .. import _lib
        L = _lib.lib()
        enc = box_preds.detach().reshape(batch_size * num_anchors, 7).contiguous()
        flat = anchors.reshape(-1, 7).contiguous().float()
        dirs, bins = None, 0
        if dir_cls_preds is not None:
            dirs = dir_cls_preds.detach().reshape(batch_size * num_anchors, -1).contiguous().float()
            bins = dirs.shape[1]
        out = torch.empty((batch_size, num_anchors, 7), dtype=torch.float32, device=box_preds.device)
        period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS if dirs is not None else 1.0
        _lib.check(L.dm_anchor_decode(_lib.ptr(enc), _lib.ptr(flat), _lib.ptr(dirs), enc.shape[0], num_anchors,
                                      bins, float(self.model_cfg.DIR_OFFSET) if dirs is not None else 0.0,
                                      float(self.model_cfg.DIR_LIMIT_OFFSET) if dirs is not None else 0.0,
                                      float(period), _lib.ptr(out), _lib.stream()), 'dm_anchor_decode')
        return out

    # ---- losses -------------------------------------------------------------
    def get_cls_layer_loss(self):
        """:101-133"""
        cls_preds = self.forward_ret_dict['cls_preds']
        box_cls_labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(cls_preds.shape[0])
        cared = box_cls_labels >= 0
        positives = box_cls_labels > 0
        negatives = box_cls_labels == 0
        cls_weights = (negatives * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(1, keepdim=True).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        cls_targets = box_cls_labels * cared.type_as(box_cls_labels)
        one_hot = torch.zeros(*list(cls_targets.shape), self.num_class + 1, dtype=cls_preds.dtype,
                              device=cls_targets.device)
        one_hot.scatter_(-1, cls_targets.unsqueeze(dim=-1).long(), 1.0)
        cls_preds = cls_preds.view(batch_size, -1, self.num_class)
        cls_loss_src = self.cls_loss_func(cls_preds, one_hot[..., 1:], weights=cls_weights)
        cls_loss = cls_loss_src.sum() / batch_size
        cls_loss = cls_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['cls_weight']
        return cls_loss, {'rpn_loss_cls': cls_loss.detach()}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        """:135-142: sin(a - b) = sin a cos b - cos a sin b"""
        rad_pred = torch.sin(boxes1[..., dim:dim + 1]) * torch.cos(boxes2[..., dim:dim + 1])
        rad_tg = torch.cos(boxes1[..., dim:dim + 1]) * torch.sin(boxes2[..., dim:dim + 1])
        boxes1 = torch.cat([boxes1[..., :dim], rad_pred, boxes1[..., dim + 1:]], dim=-1)
        boxes2 = torch.cat([boxes2[..., :dim], rad_tg, boxes2[..., dim + 1:]], dim=-1)
        return boxes1, boxes2

    @staticmethod
    def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0, num_bins=2):
        """:144-160"""
        batch_size = reg_targets.shape[0]
        anchors = anchors.view(batch_size, -1, anchors.shape[-1])
        rot_gt = reg_targets[..., 6] + anchors[..., 6]
        offset_rot = U.limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
        dir_cls_targets = torch.floor(offset_rot / (2 * np.pi / num_bins)).long()
        dir_cls_targets = torch.clamp(dir_cls_targets, min=0, max=num_bins - 1)
        if one_hot:
            t = torch.zeros(*list(dir_cls_targets.shape), num_bins, dtype=anchors.dtype,
                            device=dir_cls_targets.device)
            t.scatter_(-1, dir_cls_targets.unsqueeze(dim=-1).long(), 1.0)
            dir_cls_targets = t
        return dir_cls_targets

    def get_box_reg_layer_loss(self):
        """:162-214"""
        box_preds = self.forward_ret_dict['box_preds']
        box_dir_cls_preds = self.forward_ret_dict.get('dir_cls_preds', None)
        box_reg_targets = self.forward_ret_dict['box_reg_targets']
        box_cls_labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(box_preds.shape[0])
        positives = box_cls_labels > 0
        reg_weights = positives.float()
        pos_normalizer = positives.sum(1, keepdim=True).float()
        reg_weights = reg_weights / torch.clamp(pos_normalizer, min=1.0)
        anchors = self._cat_anchors()
        anchors = anchors.view(1, -1, anchors.shape[-1]).repeat(batch_size, 1, 1)
        box_preds = box_preds.view(batch_size, -1,
                                   box_preds.shape[-1] // self.num_anchors_per_location)
        box_preds_sin, reg_targets_sin = self.add_sin_difference(box_preds, box_reg_targets)
        loc_loss_src = self.reg_loss_func(box_preds_sin, reg_targets_sin, weights=reg_weights)
        loc_loss = loc_loss_src.sum() / batch_size
        loc_loss = loc_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['loc_weight']
        box_loss = loc_loss
        tb_dict = {'rpn_loss_loc': loc_loss.detach()}
        if box_dir_cls_preds is not None:
            dir_targets = self.get_direction_target(anchors, box_reg_targets,
                                                    dir_offset=self.model_cfg.DIR_OFFSET,
                                                    num_bins=self.model_cfg.NUM_DIR_BINS)
            dir_logits = box_dir_cls_preds.view(batch_size, -1, self.model_cfg.NUM_DIR_BINS)
            weights = positives.type_as(dir_logits)
            weights = weights / torch.clamp(weights.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.dir_loss_func(dir_logits, dir_targets, weights=weights)
            dir_loss = dir_loss.sum() / batch_size
            dir_loss = dir_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['dir_weight']
            box_loss = box_loss + dir_loss
            tb_dict['rpn_loss_dir'] = dir_loss.detach()
        return box_loss, tb_dict

    def get_loss(self):
        """anchor_head_template.py:216-223.  On the GPU the three losses come from the fused kernel
        (csrc/anchor_loss.hip); `get_loss_torch` is the element-wise restatement of the reference (the one
        the reference-generated goldens pin, and the numerics reference of the kernel)."""
        if fused_on() and self.forward_ret_dict['cls_preds'].is_cuda:
            return self.get_loss_fused()
        return self.get_loss_torch()

    def get_loss_torch(self):
        cls_loss, tb_dict = self.get_cls_layer_loss()
        box_loss, tb_dict_box = self.get_box_reg_layer_loss()
        tb_dict.update(tb_dict_box)
        rpn_loss = cls_loss + box_loss
        tb_dict['rpn_loss'] = rpn_loss.detach()
        return rpn_loss, tb_dict

    def get_loss_fused(self):
        d = self.forward_ret_dict
        cls_preds, box_preds = d['cls_preds'], d['box_preds']
        dir_preds = d.get('dir_cls_preds', None)
        labels = d['box_cls_labels']
        b = int(cls_preds.shape[0])
        a = labels.shape[1]
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        n_bins = int(self.model_cfg.NUM_DIR_BINS) if dir_preds is not None else 0
        meta = (self.num_class, n_bins, 0.25, float(self.reg_loss_func.beta),
                float(self.model_cfg.get('DIR_OFFSET', 0.0)),
                [float(lw['cls_weight']), float(lw['loc_weight']), float(lw.get('dir_weight', 0.0))],
                [float(v) for v in lw['code_weights']])
        anchors = getattr(self, '_anchors_flat', None)
        if anchors is None or anchors.device != cls_preds.device:
            anchors = self._anchors_flat = self._cat_anchors().reshape(-1, 7).contiguous()
        num_pos = (labels > 0).sum(dim=1).float()
        parts = _FusedAnchorHeadLoss.apply(
            cls_preds.reshape(b, a, self.num_class).contiguous(), box_preds.reshape(b, a, 7).contiguous(),
            dir_preds.reshape(b, a, n_bins).contiguous() if dir_preds is not None else None,
            labels.int().contiguous(), d['box_reg_targets'].contiguous(), anchors, num_pos, meta)
        cls_loss, loc_loss = parts[0], parts[1]
        tb_dict = {'rpn_loss_cls': cls_loss.detach(), 'rpn_loss_loc': loc_loss.detach()}
        rpn_loss = cls_loss + loc_loss
        if dir_preds is not None:
            rpn_loss = rpn_loss + parts[2]
            tb_dict['rpn_loss_dir'] = parts[2].detach()
        tb_dict['rpn_loss'] = rpn_loss.detach()
        return rpn_loss, tb_dict


# ------------------------------------------------------------------- point head
class _FusedPointFocalLoss(torch.autograd.Function):
    """dm_point_focal_loss (csrc/roi_targets.hip): the key-point segmentation loss and its gradient in
    one launch -> tensor [loss, #positive]."""

    @staticmethod
    def forward(ctx, preds, labels, alpha, weight):
        from .. import _lib
        _lib.require_device(preds, labels)
        n, c = int(preds.shape[0]), int(preds.shape[1])
        out = torch.empty(2, dtype=torch.float32, device=preds.device)
        grad = torch.empty_like(preds)
        _lib.check(_lib.lib().dm_point_focal_loss(_lib.ptr(preds), _lib.ptr(labels), n, c, float(alpha),
                                                  float(weight), _lib.ptr(out), _lib.ptr(grad),
                                                  _lib.stream()), 'dm_point_focal_loss')
        ctx.save_for_backward(grad)
        return out

    @staticmethod
    def backward(ctx, g):
        grad, = ctx.saved_tensors
        return grad * g[0], None, None, None


class PointHeadSimple(nn.Module):
    """point_head_template.py:9-153 + point_head_simple.py:7-91 (keypoint segmentation)."""

    def __init__(self, num_class, input_channels, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.cls_loss_func = U.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)
        self.forward_ret_dict = None
        self.cls_layers = self.make_fc_layers(model_cfg.CLS_FC, input_channels, num_class)

    @staticmethod
    def make_fc_layers(fc_cfg, input_channels, output_channels):
        layers = []
        c_in = input_channels
        for k in range(len(fc_cfg)):
            layers.extend([nn.Linear(c_in, fc_cfg[k], bias=False), nn.BatchNorm1d(fc_cfg[k]),
                           nn.ReLU()])
            c_in = fc_cfg[k]
        layers.append(nn.Linear(c_in, output_channels, bias=True))
        return nn.Sequential(*layers)

    def assign_targets(self, input_dict, fused=None):
        """point_head_simple.py:20-48 + assign_stack_targets (set_ignore_flag branch),
        batched: every sample has the same number of keypoints."""
        point_coords = input_dict['point_coords']
        gt_boxes = input_dict['gt_boxes']
        batch_size = gt_boxes.shape[0]
        if fused_on(fused) and point_coords.is_cuda and gt_boxes.shape[1] > 0:
            return {'point_cls_labels': self._assign_targets_device(point_coords, gt_boxes)}
        extend_gt_boxes = U.enlarge_box3d(gt_boxes.view(-1, gt_boxes.shape[-1]),
                                          extra_width=self.model_cfg.TARGET_CONFIG.GT_EXTRA_WIDTH
                                          ).view(batch_size, -1, gt_boxes.shape[-1])
        pts = point_coords[:, 1:4].reshape(batch_size, -1, 3)
        box_idx = roiaware_pool3d.points_in_boxes_gpu(pts, gt_boxes[:, :, 0:7]).long()
        ext_idx = roiaware_pool3d.points_in_boxes_gpu(pts, extend_gt_boxes[:, :, 0:7]).long()
        fg = box_idx >= 0
        ignore = fg ^ (ext_idx >= 0)
        if self.num_class == 1:
            fg_labels = torch.ones_like(box_idx)
        else:
            fg_labels = torch.gather(gt_boxes[:, :, -1].long(), 1, box_idx.clamp(min=0))
        labels = torch.where(ignore, torch.full_like(box_idx, -1), torch.zeros_like(box_idx))
        labels = torch.where(fg, fg_labels, labels)
        return {'point_cls_labels': labels.view(-1)}

    def _assign_targets_device(self, point_coords, gt_boxes):
        """dm_point_targets: both point-in-box tests and the label rules in one launch."""
        from .. import _lib
        pc = point_coords.detach().float().contiguous()      # (B*P, 4) [batch, x, y, z], P per sample
        gt = gt_boxes.detach().float().contiguous()
        _lib.require_device(pc, gt)
        b, g, gtc = int(gt.shape[0]), int(gt.shape[1]), int(gt.shape[2])
        p = pc.shape[0] // b
        labels = torch.empty(b * p, dtype=torch.int64, device=pc.device)
        _lib.check(_lib.lib().dm_point_targets(
            _lib.ptr(pc[:, 1:]), 4, _lib.ptr(gt), b, p, g, gtc,
            _lib.floats(self.model_cfg.TARGET_CONFIG.GT_EXTRA_WIDTH), self.num_class, _lib.ptr(labels),
            _lib.stream()), 'dm_point_targets')
        return labels

    def forward(self, batch_dict):
        if self.model_cfg.get('USE_POINT_FEATURES_BEFORE_FUSION', False):
            point_features = batch_dict['point_features_before_fusion']
        else:
            point_features = batch_dict['point_features']
        point_cls_preds = fc_rows(self.cls_layers, point_features)
        ret_dict = {'point_cls_preds': point_cls_preds}
        batch_dict['point_cls_scores'], _ = torch.sigmoid(point_cls_preds).max(dim=-1)
        if self.training:
            ret_dict['point_cls_labels'] = self.assign_targets(batch_dict)['point_cls_labels']
        self.forward_ret_dict = ret_dict
        return batch_dict

    def get_loss(self, tb_dict=None, fused=None):
        """point_head_template.py:131-154"""
        tb_dict = {} if tb_dict is None else tb_dict
        labels = self.forward_ret_dict['point_cls_labels'].view(-1)
        preds = self.forward_ret_dict['point_cls_preds'].view(-1, self.num_class)
        if fused_on(fused) and preds.is_cuda and labels.dtype == torch.int64:
            parts = _FusedPointFocalLoss.apply(
                preds.contiguous(), labels.contiguous(), 0.25,
                float(self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['point_cls_weight']))
            tb_dict.update({'point_loss_cls': parts[0].detach(), 'point_pos_num': parts[1].detach()})
            return parts[0], tb_dict
        positives = labels > 0
        cls_weights = ((labels == 0) * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(dim=0).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        one_hot = preds.new_zeros(*list(labels.shape), self.num_class + 1)
        one_hot.scatter_(-1, (labels * (labels >= 0).long()).unsqueeze(dim=-1).long(), 1.0)
        loss = self.cls_loss_func(preds, one_hot[..., 1:], weights=cls_weights).sum()
        loss = loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['point_cls_weight']
        tb_dict.update({'point_loss_cls': loss.detach(), 'point_pos_num': pos_normalizer.detach()})
        return loss, tb_dict
