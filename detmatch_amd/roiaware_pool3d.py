"""points_in_boxes_gpu — host-side mirror of
pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:28-41 (the only roiaware op PV-RCNN uses)."""
import torch

from . import _lib


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7) -> box_idxs_of_pts (B, M) int32, background = -1."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    pts = points.contiguous().float()
    bxs = boxes.contiguous().float()
    _lib.require_device(pts, bxs)
    batch_size, num_points, _ = pts.shape
    out = torch.empty((batch_size, num_points), dtype=torch.int32, device=pts.device)
    rc = _lib.lib().dm_points_in_boxes(batch_size, bxs.shape[1], num_points, _lib.ptr(bxs),
                                       _lib.ptr(pts), _lib.ptr(out), _lib.stream())
    _lib.check(rc, 'dm_points_in_boxes')
    return out
