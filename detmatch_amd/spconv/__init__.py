"""Sparse convolution — host-side mirror of mmdet3d/ops/spconv (== pcdet/ops/spconv).

Same public names as the reference package (`SparseConvTensor`, `SubMConv3d`,
`SparseConv3d`, `SparseSequential`, `ops`, `functional`); the compute lives in
libdetmatch_hip.so (rulebook.hip, spconv.hip).  Only what VoxelBackBone8x uses is
provided: SubM / strided 3-D convolution; max-pool, fused-BN, inverse/transposed
and 2-D/4-D variants are out of scope.
"""
from .conv import SparseConv3d, SparseConvolution, SubMConv3d
from .modules import SparseModule, SparseSequential
from .structure import SparseConvTensor, scatter_nd

__all__ = ['SparseConv3d', 'SubMConv3d', 'SparseConvolution', 'SparseModule',
           'SparseSequential', 'SparseConvTensor', 'scatter_nd']
