"""Hard voxelization — host-side mirror of mmdet3d/ops/voxel/voxelize.py.

Same names and argument meaning as the reference (`voxelization`, `Voxelization`
:13-58, :64-122); the compute is `dm_hard_voxelize` in libdetmatch_hip.so.
`voxelize_batch` is the batched entry the OpenPCDet adapter uses instead of the
reference's per-sample Python loop (mmdet3d/models/detectors/openpcdet.py:61-76).
"""
import ctypes

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib


def voxelize_batch(points_list, voxel_size, coors_range, max_points, max_voxels,
                   with_mean=True, sync=True):
    """Voxelize a list of (N_i, C) point tensors in one launch sequence.

    Returns (voxels (V,max_points,C), coors (V,4) int32 [b,z,y,x], num_points (V)
    int32, mean_feats (V,C) or None, voxel_counts (B+1) int32 device tensor).
    With sync=True the outputs are sliced to the total voxel count V (one D2H
    read of 4 bytes, the same read-back the reference does per sample at
    voxelization_cuda.cu:322-323); with sync=False they keep their capacity
    B*max_voxels and the caller slices later.
    """
    pts = points_list if isinstance(points_list, (list, tuple)) else [points_list]
    _lib.require_device(*pts)
    batch = len(pts)
    c = pts[0].shape[1]
    offs = [0]
    for p in pts:
        if p.dtype != torch.float32 or p.dim() != 2 or p.shape[1] != c:
            raise _lib.DetMatchHipError('points must be float32 (N, C) with equal C')
        offs.append(offs[-1] + p.shape[0])
    points = pts[0] if batch == 1 else torch.cat(pts, dim=0)
    dev = points.device
    n = offs[-1]
    cap = batch * max_voxels
    voxels = torch.empty((cap, max_points, c), dtype=torch.float32, device=dev)
    coors = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, c), dtype=torch.float32, device=dev) if with_mean else None
    counts = torch.empty((batch + 1,), dtype=torch.int32, device=dev)
    L = _lib.lib()
    wsb = L.dm_hard_voxelize_workspace_bytes(n, batch)
    ws = _lib.workspace(wsb, dev, 'voxelize')
    offs_c = (ctypes.c_int32 * (batch + 1))(*offs)
    rc = L.dm_hard_voxelize(_lib.ptr(points), n, c, offs_c, batch, _lib.floats(voxel_size),
                            _lib.floats(coors_range), int(max_points), int(max_voxels), 4,
                            _lib.ptr(voxels), _lib.ptr(coors), _lib.ptr(num), _lib.ptr(mean),
                            _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, 'dm_hard_voxelize')
    if sync:
        v = int(counts[batch].item())
        voxels, coors, num = voxels[:v], coors[:v], num[:v]
        if mean is not None:
            mean = mean[:v]
    return voxels, coors, num, mean, counts


class _Voxelization(torch.autograd.Function):
    """mmdet3d/ops/voxel/voxelize.py:11-58 (hard voxelization branch)."""

    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000):
        if max_points == -1 or max_voxels == -1:
            raise NotImplementedError(
                'dynamic voxelization is off the DetMatch hot path (SURVEY §2.1)')
        voxels, coors, num, _, _ = voxelize_batch([points.contiguous()], voxel_size, coors_range,
                                                  max_points, max_voxels, with_mean=False)
        return voxels, coors[:, 1:].contiguous(), num


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    """mmdet3d/ops/voxel/voxelize.py:64-122."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        if isinstance(max_voxels, tuple):
            self.max_voxels = max_voxels
        else:
            self.max_voxels = _pair(max_voxels)
        point_cloud_range = torch.tensor(point_cloud_range, dtype=torch.float32)
        voxel_size = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = torch.round(grid_size).long()
        input_feat_shape = grid_size[:2]
        self.grid_size = grid_size
        self.pcd_shape = [*input_feat_shape, 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range,
                            self.max_num_points, max_voxels)

    def __repr__(self):
        return (self.__class__.__name__ + '(voxel_size=' + str(self.voxel_size) +
                ', point_cloud_range=' + str(self.point_cloud_range) + ', max_num_points=' +
                str(self.max_num_points) + ', max_voxels=' + str(self.max_voxels) + ')')
