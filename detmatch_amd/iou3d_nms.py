"""Rotated 3D IoU + NMS — host-side mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py.

Same function names and argument meaning (:31-110); compute in libdetmatch_hip.so
(iou3d_nms.hip).  Unlike the reference nothing is copied to the host inside the ops:
`keep` and `num_keep` stay on the device; the only read-back is the 4-byte keep count
needed to size the returned index tensor.
"""
import torch

from . import _lib


def _check(boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a = boxes_a.contiguous().float()
    b = boxes_b.contiguous().float()
    _lib.require_device(a, b)
    return a, b


def _pair(fn_name, boxes_a, boxes_b):
    a, b = _check(boxes_a, boxes_b)
    L = _lib.lib()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    ws = _lib.workspace(L.dm_iou3d_workspace_bytes(a.shape[0], b.shape[0]), a.device, 'iou3d')
    rc = getattr(L, fn_name)(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _lib.ptr(out),
                             _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, fn_name)
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N,7),(M,7) [x,y,z,dx,dy,dz,heading] -> rotated BEV overlap AREA (N,M)."""
    return _pair('dm_boxes_overlap_bev', boxes_a, boxes_b)


def boxes_overlap_bev_exact(boxes_a, boxes_b):
    """Rotated BEV overlap AREA (N, M) by exact convex clipping in double precision — no corner margin.
    For comparisons with thresholds / zero (KITTI evaluation, GT-paste collision test)."""
    a, b = _check(boxes_a, boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    rc = _lib.lib().dm_boxes_overlap_bev_exact(_lib.ptr(a), a.shape[0], _lib.ptr(b), b.shape[0], _lib.ptr(out),
                                               _lib.stream())
    _lib.check(rc, 'dm_boxes_overlap_bev_exact')
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:31-45"""
    return _pair('dm_boxes_iou_bev', boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:48-81: BEV overlap x height overlap / union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a, boxes_b)
    max_of_min = torch.max(a_min, b_min)
    min_of_max = torch.min(a_max, b_max)
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms(fn_name, boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    assert boxes.shape[1] == 7
    # stable: equal scores keep their original relative order (the reference's sort is
    # not stable, SURVEY K4 (i) — this build fixes the rule "lower index first")
    order = _lib.sort_rows(scores, descending=True)
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous().float()
    _lib.require_device(b)
    n = b.shape[0]
    L = _lib.lib()
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    num = torch.zeros((1,), dtype=torch.int32, device=b.device)
    ws = _lib.workspace(L.dm_nms_workspace_bytes(n), b.device, 'nms')
    rc = getattr(L, fn_name)(_lib.ptr(b), n, float(thresh),
                             int(post_max_size) if post_max_size else 0, _lib.ptr(keep),
                             _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, fn_name)
    k = int(num.item())
    return order[keep[:k]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None, **kwargs):
    """iou3d_nms_utils.py:84-104 -> (kept indices into `boxes`, None).

    post_max_size (extension): stop the greedy pass after that many survivors."""
    return _nms('dm_nms', boxes, scores, thresh, pre_maxsize, post_max_size)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:107-121 (axis-aligned IoU)."""
    return _nms('dm_nms_normal', boxes, scores, thresh, None, kwargs.get('post_max_size'))
