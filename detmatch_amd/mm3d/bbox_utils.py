"""SSL bbox utilities — mmdet3d/models/ssl_modules/bbox_utils.py (apply_3d_transformation_bboxes
:110, filter_by_nms_2d :282, modified_multiclass_nms :14, bbox_3d_to_bbox_2d :372),
mmdet3d/models/fusion_layers/coord_transform.py (extract_2d_info :91, bbox_2d_transform :121)
and mmdet3d/models/ssl_modules/utils.py (mlvl_get / mlvl_set / mlvl_getattr)."""
from functools import partial

import numpy as np
import torch

from ..devconst import const, upload

from .. import _lib
from .box3d import LiDARInstance3DBoxes


# ------------------------------------------------------------------ dotted-key access
def mlvl_get(d, key, default=None):
    key_split = key.split('.', maxsplit=1)
    if len(key_split) == 1:
        return d.get(key, default)
    cur, rest = key_split
    return mlvl_get(d[cur], rest, default) if cur in d else default


def mlvl_set(d, key, val):
    key_split = key.split('.', maxsplit=1)
    if len(key_split) == 1:
        if key in d:
            raise Exception('Key already exists')
        d[key] = val
        return
    cur, rest = key_split
    if cur not in d:
        d[cur] = dict()
    mlvl_set(d[cur], rest, val)


def mlvl_getattr(c, key, default=None):
    key_split = key.split('.', maxsplit=1)
    if len(key_split) == 1:
        return getattr(c, key)
    cur, rest = key_split
    return mlvl_getattr(getattr(c, cur), rest, default) if hasattr(c, cur) else default


# ------------------------------------------------------------------ 3D augmentation replay
def apply_3d_transformation_bboxes(bbox, img_meta, reverse=False):
    """bbox_utils.py:110-200: replay (or undo) transformation_3d_flow in {T,S,R,HF,VF}.
    The reverse rotation uses torch.inverse(M) in fp32 (:176), not M^T."""
    assert isinstance(img_meta, dict)
    dtype, device = bbox.tensor.dtype, bbox.tensor.device
    # the record of the augmentation is host data: invert / negate it on the host (torch CPU, fp32,
    # the same LAPACK path the reference takes on CPU tensors) and upload without a stream sync
    rot_h = (torch.as_tensor(np.asarray(img_meta['pcd_rotation']), dtype=dtype)
             if 'pcd_rotation' in img_meta else torch.eye(3, dtype=dtype))
    scale = img_meta['pcd_scale_factor'] if 'pcd_scale_factor' in img_meta else 1.
    trans_h = (torch.as_tensor(np.asarray(img_meta['pcd_trans']), dtype=dtype)
               if 'pcd_trans' in img_meta else torch.zeros((3), dtype=dtype))
    if isinstance(rot_h, torch.Tensor) and rot_h.is_cuda:
        rot_h, trans_h = rot_h.cpu(), trans_h.cpu()
    hflip = img_meta.get('pcd_horizontal_flip', False)
    vflip = img_meta.get('pcd_vertical_flip', False)
    flow = img_meta.get('transformation_3d_flow', [])
    bbox = bbox.clone()
    h_func = partial(bbox.flip, bev_direction='horizontal') if hflip else (lambda: None)
    v_func = partial(bbox.flip, bev_direction='vertical') if vflip else (lambda: None)
    if reverse:
        scale_func = partial(bbox.scale, scale_factor=1.0 / scale)
        translate_func = partial(bbox.translate, trans_vector=upload(-trans_h, device, dtype))
        rotate_func = partial(bbox.rotate, angle=upload(rot_h.inverse(), device, dtype))
        flow = flow[::-1]
    else:
        scale_func = partial(bbox.scale, scale_factor=scale)
        translate_func = partial(bbox.translate, trans_vector=upload(trans_h, device, dtype))
        rotate_func = partial(bbox.rotate, angle=upload(rot_h, device, dtype))
    mapping = {'T': translate_func, 'S': scale_func, 'R': rotate_func, 'HF': h_func, 'VF': v_func}
    for op in flow:
        assert op in mapping, 'This 3D data transformation op (%s) is not supported' % op
        mapping[op]()
    return bbox


def take(t, idx):
    """t[idx] for a 1-D integer index tensor (or boolean mask) over dim 0, as index_select: the same
    values, but the backward is zeros + index_add (2 launches) instead of the sort-based
    index_put(accumulate=True) of advanced indexing (~8 launches) — the selections of the pseudo-label
    glue carry gradients of a few dozen boxes."""
    if not isinstance(t, torch.Tensor):        # box containers index their tensor the same way
        return t[idx]
    if idx.dtype == torch.bool:
        idx = idx.nonzero(as_tuple=False).squeeze(1)
    return t.index_select(0, idx)


def compose_3d_transformation(img_meta, reverse=False):
    """The recorded 3D augmentations of a sample (what apply_3d_transformation_bboxes replays op by op)
    as ONE affine map of the box parameters: centre' = centre @ A + t, size' = s * size,
    yaw' = sigma * yaw + off.  Every op of transformation_3d_flow is affine in (centre, size, yaw):
    T adds to the centre, S scales centre and size, R multiplies the centre by M and adds
    atan2(M[1, 0], M[0, 0]) to the yaw, HF / VF negate y / x and reflect the yaw.  Composed in float64
    from the same fp32 records (the inverse rotation is torch.inverse in fp32, as in the replay).
    -> (A (3, 3), t (3,), s, sigma, off) as float64 numpy / floats."""
    rot = (torch.as_tensor(np.asarray(img_meta['pcd_rotation']), dtype=torch.float32).cpu()
           if 'pcd_rotation' in img_meta else torch.eye(3))
    scale = float(img_meta['pcd_scale_factor']) if 'pcd_scale_factor' in img_meta else 1.0
    trans = (torch.as_tensor(np.asarray(img_meta['pcd_trans']), dtype=torch.float32).cpu()
             if 'pcd_trans' in img_meta else torch.zeros(3))
    flow = list(img_meta.get('transformation_3d_flow', []))
    if reverse:
        scale, trans, rot, flow = 1.0 / scale, -trans, rot.inverse(), flow[::-1]
    rot = rot.double().numpy()
    trans = trans.double().numpy()
    A, t, s, sigma, off = np.eye(3), np.zeros(3), 1.0, 1.0, 0.0
    for op in flow:
        if op == 'T':
            t = t + trans
        elif op == 'S':
            A, t, s = A * scale, t * scale, s * scale
        elif op == 'R':
            A, t = A @ rot, t @ rot
            off = off + float(np.arctan2(np.float32(rot[1, 0]), np.float32(rot[0, 0])))
        elif op == 'HF':
            if img_meta.get('pcd_horizontal_flip', False):
                A, t = A.copy(), t.copy()
                A[:, 1], t[1] = -A[:, 1], -t[1]
                sigma, off = -sigma, -off + np.pi
        elif op == 'VF':
            if img_meta.get('pcd_vertical_flip', False):
                A, t = A.copy(), t.copy()
                A[:, 0], t[0] = -A[:, 0], -t[0]
                sigma, off = -sigma, -off
        else:
            raise AssertionError('This 3D data transformation op (%s) is not supported' % op)
    return A, t, s, sigma, off


class _UnaugProject(torch.autograd.Function):
    """boxes3d (N, 7) of the augmented frame -> xyxy (N, 4) in the original image + validity, one launch
    forward and one backward (csrc/box_project.hip)."""

    @staticmethod
    def forward(ctx, boxes, xf, m16, img_w, img_h):
        from .. import _lib
        boxes = boxes.detach().float().contiguous()
        n = boxes.shape[0]
        out = torch.empty((n, 4), dtype=torch.float32, device=boxes.device)
        valid = torch.empty((n,), dtype=torch.uint8, device=boxes.device)
        _lib.check(_lib.lib().dm_box3d_project_forward(_lib.ptr(boxes), n, xf, m16, float(img_w), float(img_h),
                                                       _lib.ptr(out), _lib.ptr(valid), _lib.stream()),
                   'dm_box3d_project_forward')
        ctx.save_for_backward(boxes)
        ctx.cfg = (xf, m16, float(img_w), float(img_h))
        valid = valid.bool()
        ctx.mark_non_differentiable(valid)
        return out, valid

    @staticmethod
    def backward(ctx, gout, _gvalid):
        from .. import _lib
        boxes, = ctx.saved_tensors
        xf, m16, img_w, img_h = ctx.cfg
        n = boxes.shape[0]
        g = torch.empty_like(boxes)
        _lib.check(_lib.lib().dm_box3d_project_backward(_lib.ptr(boxes), n, xf, m16, img_w, img_h,
                                                        _lib.ptr(gout.contiguous().float()), _lib.ptr(g),
                                                        _lib.stream()), 'dm_box3d_project_backward')
        return g, None, None, None, None


def unaug_project_boxes(bboxes_3d, img_meta):
    """apply_3d_transformation_bboxes(reverse=True) + bbox_3d_to_bbox_2d on the device kernels:
    (xyxy (N, 4) in the original image, valid (N,) bool); gradient flows to bboxes_3d.tensor."""
    from .. import _lib
    assert isinstance(bboxes_3d, LiDARInstance3DBoxes) and bboxes_3d.tensor.shape[1] == 7
    A, t, s, sigma, off = compose_3d_transformation(img_meta, reverse=True)
    xf = _lib.floats(list(A.reshape(-1)) + list(t) + [s, sigma, off, 0.0, 0.0])
    l2i = img_meta['lidar2img']
    m16 = _lib.floats(np.asarray(l2i.cpu() if torch.is_tensor(l2i) else l2i, np.float32).reshape(-1))
    img_v, img_h = img_meta['ori_shape'][0], img_meta['ori_shape'][1]
    return _UnaugProject.apply(bboxes_3d.tensor, xf, m16, img_h, img_v)


# ------------------------------------------------------------------ 3D -> 2D projection
def bbox_3d_to_bbox_2d(bboxes_3d, lidar2img, img_shape):
    """bbox_utils.py:372-441 (autograd preserving).  Returns xyxy (N,4) for ALL boxes and the
    validity mask: >= 3 corners inside the image and mean (clamped) depth >= 0.5."""
    assert isinstance(bboxes_3d, LiDARInstance3DBoxes)
    if not (isinstance(lidar2img, torch.Tensor) and lidar2img.is_cuda):
        lidar2img = upload(lidar2img, bboxes_3d.tensor.device, bboxes_3d.tensor.dtype)
    lidar2img = lidar2img.to(bboxes_3d.tensor)
    assert lidar2img.shape == (4, 4)
    n = len(bboxes_3d.tensor)
    img_v, img_h = img_shape[0], img_shape[1]
    if n == 0:
        return (torch.empty((0, 4), dtype=bboxes_3d.tensor.dtype, device=bboxes_3d.tensor.device),
                torch.empty((0,), dtype=torch.bool, device=bboxes_3d.tensor.device))
    corners = bboxes_3d.corners.reshape(-1, 3)
    # [corner, 1] @ lidar2img^T as three scaled columns + the translation column (no vendor GEMM for a 4-wide contraction)
    p = corners[:, 0:1] * lidar2img[:, 0] + corners[:, 1:2] * lidar2img[:, 1] + corners[:, 2:3] * lidar2img[:, 2] \
        + lidar2img[:, 3]
    depth = torch.clamp(p[:, 2], min=1e-5)     # clamp BEFORE the divide and the tests (:408)
    x = p[:, 0] / depth
    y = p[:, 1] / depth
    valid = (x >= 0) & (x < img_h) & (y >= 0) & (y < img_v) & (depth > 0)
    x, y, depth, valid = x.reshape(-1, 8), y.reshape(-1, 8), depth.reshape(-1, 8), valid.reshape(-1, 8)
    final_valid = (valid.sum(dim=1) >= 3) & (depth.mean(dim=1) >= 0.5)
    xmin = torch.clip(x.min(dim=1)[0], 0, img_h)
    ymin = torch.clip(y.min(dim=1)[0], 0, img_v)
    xmax = torch.clip(x.max(dim=1)[0], 0, img_h)
    ymax = torch.clip(y.max(dim=1)[0], 0, img_v)
    return torch.stack([xmin, ymin, xmax, ymax], dim=1), final_valid


# ------------------------------------------------------------------ 2D augmentation replay
def extract_2d_info(img_meta, tensor):
    """coord_transform.py:91-118"""
    img_h, img_w, _ = img_meta['img_shape']
    ori_h, ori_w, _ = img_meta['ori_shape']
    scale = (upload(np.asarray(img_meta['scale_factor'][:2]), tensor.device, tensor.dtype)
             if 'scale_factor' in img_meta else const([1.0, 1.0], tensor.device, tensor.dtype))
    flip = img_meta['flip'] if 'flip' in img_meta else False
    crop = (upload(np.asarray(img_meta['img_crop_offset']), tensor.device, tensor.dtype)
            if 'img_crop_offset' in img_meta else const([0.0, 0.0], tensor.device, tensor.dtype))
    return img_h, img_w, ori_h, ori_w, scale, flip, crop


class _Box2DTransform(torch.autograd.Function):
    """bbox_2d_transform on (n, 4) boxes in one launch, its transpose in one more (csrc/box_project.hip)."""

    @staticmethod
    def forward(ctx, boxes, cfg):
        ctx.cfg = cfg
        return _Box2DTransform._run(boxes.detach(), cfg, 0)

    @staticmethod
    def backward(ctx, g):
        return _Box2DTransform._run(g, ctx.cfg, 1), None

    @staticmethod
    def _run(t, cfg, backward):
        t = t.float().contiguous()
        out = torch.empty_like(t)
        sx, sy, cx, cy, img_w, flip, ori2new = cfg
        _lib.check(_lib.lib().dm_bbox2d_transform(_lib.ptr(t), t.shape[0], sx, sy, cx, cy, img_w, flip, ori2new,
                                                  backward, _lib.ptr(out), _lib.stream()), 'dm_bbox2d_transform')
        return out


def bbox_2d_transform(img_meta, bbox_2d, ori2new):
    """coord_transform.py:121-175 (scale -> crop offset -> h-flip, or the reverse)."""
    from ..fused import on as fused_on
    if fused_on() and bbox_2d.is_cuda and bbox_2d.dim() == 2 and bbox_2d.shape[1] == 4 and \
            bbox_2d.dtype == torch.float32 and bbox_2d.shape[0] > 0:
        sf = np.asarray(img_meta['scale_factor'][:2], np.float32) if 'scale_factor' in img_meta \
            else np.ones(2, np.float32)
        co = np.asarray(img_meta['img_crop_offset'], np.float32) if 'img_crop_offset' in img_meta \
            else np.zeros(2, np.float32)
        cfg = (float(sf[0]), float(sf[1]), float(co[0]), float(co[1]), float(img_meta['img_shape'][1]),
               int(bool(img_meta['flip'] if 'flip' in img_meta else False)), int(bool(ori2new)))
        return _Box2DTransform.apply(bbox_2d, cfg)
    img_h, img_w, ori_h, ori_w, scale, flip, crop = extract_2d_info(img_meta, bbox_2d)
    x1, y1, x2, y2 = bbox_2d[:, 0], bbox_2d[:, 1], bbox_2d[:, 2], bbox_2d[:, 3]
    if ori2new:
        x1, x2 = x1 * scale[0] + crop[0], x2 * scale[0] + crop[0]
        y1, y2 = y1 * scale[1] + crop[1], y2 * scale[1] + crop[1]
        if flip:
            x1, x2 = img_w - x2, img_w - x1
    else:
        if flip:
            x1, x2 = img_w - x2, img_w - x1
        x1, x2 = (x1 - crop[0]) / scale[0], (x2 - crop[0]) / scale[0]
        y1, y2 = (y1 - crop[1]) / scale[1], (y2 - crop[1]) / scale[1]
    return torch.cat([torch.stack([x1, y1, x2, y2], dim=1), bbox_2d[:, 4:]], dim=1)


# ------------------------------------------------------------------ 2D NMS
def nms_2d(boxes, scores, iou_threshold, max_num=-1):
    """mmcv.ops.nms on the device: -> (dets (k,5) sorted by score, keep indices (k,))."""
    if boxes.shape[0] == 0:
        return torch.cat([boxes, scores[:, None]], -1), boxes.new_zeros((0,), dtype=torch.long)
    order = _lib.sort_rows(scores, descending=True)
    b = boxes[order].contiguous().float()
    _lib.require_device(b)
    n = b.shape[0]
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.int64, device=b.device)
    num = torch.zeros((1,), dtype=torch.int32, device=b.device)
    ws = _lib.workspace(L.dm_nms_workspace_bytes(n), b.device, 'nms')
    _lib.check(L.dm_nms_2d(_lib.ptr(b), n, float(iou_threshold), int(max_num) if max_num > 0 else 0,
                           _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream()),
               'dm_nms_2d')
    k = int(num.item())
    inds = order[keep[:k]]
    return torch.cat([take(boxes, inds), take(scores, inds)[:, None]], -1), inds


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.nms.batched_nms (mmcv-full 1.3.16): offset boxes by idx * (max_coord + 1) so
    that classes never overlap, then plain NMS.  Extra keys of nms_cfg (e.g. max_num) ride
    along ignored, as in the reference call (bbox_utils.py:326-334)."""
    cfg = dict(nms_cfg)
    thr = cfg.get('iou_threshold', cfg.get('iou_thr', 0.5))
    if class_agnostic or boxes.shape[0] == 0:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)
        boxes_for_nms = boxes + offsets[:, None]
    dets, keep = nms_2d(boxes_for_nms, scores, thr)
    return torch.cat([take(boxes, keep), dets[:, -1:]], -1), keep


def modified_multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """bbox_utils.py:14-108 with fixed_return_inds=True: -> (dets (k,5), labels (k), flat
    (box x class) indices w.r.t. the tensors BEFORE the score threshold)."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device)
    labels = labels.view(1, -1).expand_as(scores)
    bboxes = bboxes.reshape(-1, 4)
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    valid_mask = scores > score_thr
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = take(bboxes, inds), take(scores, inds), take(labels, inds)
    if bboxes.numel() == 0:
        return torch.cat([bboxes, scores[:, None]], -1), labels, inds
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets = dets[:max_num]
        keep = keep[:max_num]
    return dets, take(labels, keep), take(inds, keep)


def filter_by_nms_2d(bbox_list, nms_cfg, use_sigmoid_cls, return_indices=False):
    """bbox_utils.py:282-369: per-class 2D NMS keeping each survivor's full score vector."""
    assert not return_indices
    res = []
    for bboxes, scores in bbox_list:
        assert bboxes.shape[0] == scores.shape[0] and len(scores.shape) == 2
        nms_pre = nms_cfg.get('nms_pre', -1)
        if nms_pre > 0 and scores.shape[0] > nms_pre:
            max_scores = scores.max(dim=1)[0] if use_sigmoid_cls else scores[:, :-1].max(dim=1)[0]
            _, topk_inds = max_scores.topk(nms_pre)
            bboxes, scores = take(bboxes, topk_inds), take(scores, topk_inds)
        if use_sigmoid_cls:
            scores_for_nms = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
        else:
            scores_for_nms = scores
        cfg = {k: v for k, v in dict(nms_cfg).items() if k not in ('score_thr', 'nms_pre')}
        score_thr = nms_cfg.get('score_thr', None)
        dets, _, selected = modified_multiclass_nms(bboxes, scores_for_nms,
                                                    score_thr if score_thr is not None else 0, cfg,
                                                    nms_cfg.get('max_num', -1))
        sel_box = selected.long() // (scores_for_nms.shape[1] - 1)
        res.append((dets[:, :4], take(scores, sel_box)))
    return res


def filter_by_nms_2d_masked(entries, nms_cfg, use_sigmoid_cls):
    """filter_by_nms_2d without a single device->host read: entries = [((boxes (n, 4), scores (n, C[+1])), keep | None)],
    `keep` an optional bool mask over the n rows (rows a filter in front rejected).  -> [((boxes (K, 4), scores (K, .)),
    valid (K,) bool)] with K = min(max_num, n * C) FIXED rows of which the first valid.sum() are, in order, exactly the
    rows filter_by_nms_2d returns on the compacted list:
      * the score threshold and `keep` do not select candidates, they push them behind every valid candidate in the
        score order (stable sort of the same keys: valid candidates keep their relative order), where the greedy pass
        can neither let them suppress a valid box nor keep one of them before the last valid survivor;
      * the class offset of batched_nms is (max coordinate over the VALID candidates + 1), as on the compacted list;
      * the greedy pass stops after max_num survivors (== dets[:max_num] of modified_multiclass_nms).
    Needs nms_pre <= 0 (the configs' value) and max_num > 0."""
    res = []
    L = _lib.lib()
    thr = float(nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr', 0.5)))
    score_thr = nms_cfg.get('score_thr', None)
    score_thr = 0 if score_thr is None else score_thr
    max_num = int(nms_cfg.get('max_num', -1))
    assert nms_cfg.get('nms_pre', -1) <= 0
    for (bboxes, scores), keep_in in entries:
        assert bboxes.shape[0] == scores.shape[0] and scores.dim() == 2
        n = bboxes.shape[0]
        c = scores.shape[1] if use_sigmoid_cls else scores.shape[1] - 1
        assert bboxes.shape[1] in (4, 4 * c)          # class-agnostic or one box per class (modified_multiclass_nms :35-39)
        if n == 0 or c == 0:
            res.append(((bboxes, scores), None))
            continue
        dev = bboxes.device
        flat_s = scores[:, :c].reshape(-1)                              # candidate (box, class) = box * c + class
        valid = flat_s > score_thr
        if keep_in is not None:
            valid = valid & keep_in[:, None].expand(n, c).reshape(-1)
        flat_b = bboxes.view(n, c, 4).reshape(-1, 4) if bboxes.shape[1] > 4 else bboxes[:, None, :].expand(n, c, 4).reshape(-1, 4)
        labels = torch.arange(c, device=dev, dtype=bboxes.dtype).repeat(n)
        neg = torch.full_like(flat_s, float('-inf'))
        max_coord = torch.where(valid[:, None], flat_b, neg[:, None]).max()
        key = torch.where(valid, flat_s, neg)
        order = _lib.sort_rows(key, descending=True)
        sorted_boxes = (flat_b + (labels * (max_coord + 1))[:, None])[order].contiguous().float()
        m = n * c
        k_rows = min(max_num, m) if max_num > 0 else m
        keep = torch.empty((m,), dtype=torch.int64, device=dev)
        num = torch.zeros((1,), dtype=torch.int32, device=dev)
        ws = _lib.workspace(L.dm_nms_workspace_bytes(m), dev, 'nms')
        _lib.check(L.dm_nms_2d(_lib.ptr(sorted_boxes), m, thr, k_rows if max_num > 0 else 0, _lib.ptr(keep), _lib.ptr(num),
                               _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_nms_2d')
        slot = torch.arange(k_rows, device=dev)
        pos = torch.where(slot < num, keep[:k_rows], torch.zeros_like(slot))     # positions in the score order
        ok = (slot < num) & (pos < valid.sum())                                  # (valid candidates come first in it)
        cand = order.index_select(0, pos)
        res.append(((take(flat_b, cand), take(scores, cand // c)), ok))
    return res


# ------------------------------------------------------------------ 3D multi-class NMS
def xywhr2xyxyr(boxes_xywhr):
    """mmdet3d/core/bbox/structures/utils.py:62-82: BEV (cx, cy, w, h, r) -> (x1, y1, x2, y2, r)."""
    half = boxes_xywhr[:, 2:4] / 2
    return torch.cat([boxes_xywhr[:, 0:2] - half, boxes_xywhr[:, 0:2] + half, boxes_xywhr[:, 4:5]], dim=1)


def _xyxyr_as_boxes7(b):
    """(x1,y1,x2,y2,r) rotated BEV rectangles as the 7-value boxes the device NMS takes.  mmdet3d's
    iou3d kernel turns its corners CLOCKWISE by r (mmdet3d/ops/iou3d/src/iou3d_kernel.cu:111-118), the
    device NMS counter-clockwise by the heading: heading = -r."""
    z = torch.zeros_like(b[:, :1])
    return torch.cat([(b[:, 0:1] + b[:, 2:3]) / 2, (b[:, 1:2] + b[:, 3:4]) / 2, z, b[:, 2:3] - b[:, 0:1],
                      b[:, 3:4] - b[:, 1:2], z + 1, -b[:, 4:5]], dim=1)


def box3d_multiclass_nms(mlvl_bboxes, mlvl_bboxes_for_nms, mlvl_scores, score_thr, max_num, cfg,
                         mlvl_attr_scores=None):
    """mmdet3d/core/post_processing/box3d_nms.py:9-120: per-class rotated (or axis-aligned) BEV NMS on
    (x1,y1,x2,y2,r) boxes, classes concatenated, top `max_num` by score
    -> (bboxes, scores, labels[, attr_scores])."""
    from .. import iou3d_nms
    num_classes = mlvl_scores.shape[1] - 1
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    nms = iou3d_nms.nms_gpu if get('use_rotate_nms') else iou3d_nms.nms_normal_gpu
    bboxes, scores, labels, attrs = [], [], [], []
    for i in range(num_classes):
        cls_inds = mlvl_scores[:, i] > score_thr
        if not bool(cls_inds.any()):
            continue
        _scores = mlvl_scores[cls_inds, i]
        selected, _ = nms(_xyxyr_as_boxes7(mlvl_bboxes_for_nms[cls_inds, :]), _scores, get('nms_thr'))
        bboxes.append(mlvl_bboxes[cls_inds, :][selected])
        scores.append(_scores[selected])
        labels.append(torch.full((len(selected),), i, dtype=torch.long, device=mlvl_bboxes.device))
        if mlvl_attr_scores is not None:
            attrs.append(mlvl_attr_scores[cls_inds][selected])
    if bboxes:
        bboxes, scores, labels = torch.cat(bboxes, 0), torch.cat(scores, 0), torch.cat(labels, 0)
        attrs = torch.cat(attrs, 0) if mlvl_attr_scores is not None else None
        if bboxes.shape[0] > max_num:
            inds = scores.sort(descending=True)[1][:max_num]
            bboxes, scores, labels = bboxes[inds, :], scores[inds], labels[inds]
            attrs = attrs[inds] if attrs is not None else None
    else:
        bboxes = mlvl_scores.new_zeros((0, mlvl_bboxes.size(-1)))
        scores = mlvl_scores.new_zeros((0,))
        labels = mlvl_scores.new_zeros((0,), dtype=torch.long)
        attrs = mlvl_scores.new_zeros((0,)) if mlvl_attr_scores is not None else None
    return (bboxes, scores, labels) + ((attrs,) if mlvl_attr_scores is not None else ())


def filter_by_nms(raw_bbox_list, nms_cfg, use_sigmoid_cls, return_labels=False):
    """bbox_utils.py:203-279: multi-class 3D NMS of (boxes, (n, C) score matrix) pairs that keeps each
    survivor's FULL score vector (a survivor's class label need not be its arg-max)."""
    res = []
    get = nms_cfg.get if isinstance(nms_cfg, dict) else (lambda k, d=None: getattr(nms_cfg, k, d))
    for bboxes, scores in raw_bbox_list:
        assert bboxes.tensor.shape[0] == scores.shape[0] and scores.dim() == 2
        pred = bboxes.tensor
        nms_pre = get('nms_pre', -1)
        if nms_pre > 0 and scores.shape[0] > nms_pre:
            max_scores = scores.max(dim=1)[0] if use_sigmoid_cls else scores[:, :-1].max(dim=1)[0]
            topk = max_scores.topk(nms_pre)[1]
            pred, scores = pred[topk, :], scores[topk, :]
        for_nms = xywhr2xyxyr(type(bboxes)(pred, box_dim=pred.shape[-1]).bev)
        scores_for_nms = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1) if use_sigmoid_cls \
            else scores
        idx = torch.arange(len(pred), dtype=torch.float32, device=pred.device)
        sel_boxes, _, sel_labels, sel_idx = box3d_multiclass_nms(pred, for_nms, scores_for_nms, get('score_thr', 0),
                                                                 get('max_num'), nms_cfg, mlvl_attr_scores=idx)
        sel_idx = sel_idx.long()
        out = (type(bboxes)(sel_boxes, box_dim=sel_boxes.shape[-1]), scores[sel_idx])
        res.append(out + ((sel_labels,) if return_labels else ()))
    return res
