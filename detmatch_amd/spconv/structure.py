"""Sparse tensor container of the spconv API surface (the reference: mmdet3d/ops/spconv/structure.py —
`SparseConvTensor` :21-69, `scatter_nd` :5-18).  Same attributes and methods, because the modules and
the PV-RCNN code address them by name; densification goes through one flat row copy."""
import numpy as np
import torch


def _linear_rows(indices, shape):
    """Row-major linear index of integer coordinates (N, k) in a grid of size shape[:k]."""
    lin = indices[:, 0].long()
    for d in range(1, indices.shape[1]):
        lin = lin * int(shape[d]) + indices[:, d].long()
    return lin


def scatter_nd(indices, updates, shape):
    """out[tuple(indices[i])] = updates[i] on zeros(shape); duplicates are not accumulated (the last
    writer wins, as in the reference)."""
    k = indices.shape[-1]
    idx = indices.reshape(-1, k)
    rows = int(np.prod(shape[:k]))
    tail = [int(s) for s in shape[k:]]
    flat = torch.zeros([rows] + tail, dtype=updates.dtype, device=updates.device)
    flat[_linear_rows(idx, shape)] = updates.reshape([idx.shape[0]] + tail)
    return flat.view(*[int(s) for s in shape])


class SparseConvTensor(object):
    """features (N, C) fp32 rows, indices (N, 4) int32 [batch, z, y, x] of the active cells of a
    `spatial_shape` grid; `indice_dict` caches the rulebooks of the layers that share an indice_key."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.grid = grid
        self.indice_dict = {}

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    @property
    def sparity(self):
        return self.indices.shape[0] / float(self.spatial_size) / self.batch_size

    def dense(self, channels_first=True):
        """(B, C, *spatial) — or (B, *spatial, C) with channels_first=False — with zeros at inactive cells."""
        grid = [self.batch_size] + [int(s) for s in self.spatial_shape]
        out = scatter_nd(self.indices, self.features, grid + [self.features.shape[1]])
        if channels_first:
            last = out.dim() - 1
            out = out.permute(0, last, *range(1, last)).contiguous()
        return out
