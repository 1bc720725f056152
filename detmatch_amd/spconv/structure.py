"""mmdet3d/ops/spconv/structure.py:5-69 — SparseConvTensor container."""
import numpy as np
import torch


def scatter_nd(indices, updates, shape):
    """structure.py:5-18: out[indices] = updates on a zero tensor of `shape`."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    output_shape = list(indices.shape[:-1]) + shape[indices.shape[-1]:]
    flat = indices.view(-1, ndim)
    slices = [flat[:, i] for i in range(ndim)] + [Ellipsis]
    ret[tuple(slices)] = updates.view(*output_shape)
    return ret


class SparseConvTensor(object):
    """features (N, C) f32, indices (N, 4) int32 [b, z, y, x]."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices = self.indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        """structure.py:55-64: scatter to (B, [C,] D, H, W[, C])."""
        output_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.long(), self.features, output_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        perm = list(range(0, ndim + 1))
        perm.insert(1, ndim + 1)
        return res.permute(*perm).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
