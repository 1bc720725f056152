"""Sparse convolution modules with the reference's public surface (mmdet3d/ops/spconv/conv.py:
`SparseConvolution`, `SparseConv3d`, `SubMConv3d` — same constructor arguments, attribute names,
parameter names and the (kz, ky, kx, Cin, Cout) weight layout of conv.py:98-99, so released
checkpoints load unchanged) computed by this repo's kernels: rulebook.hip builds the gather tables
once per `indice_key`, spconv.hip runs the fused gather-GEMM.

Only what VoxelBackBone8x uses is implemented (3-D, groups 1, dilation 1, no transposed / inverse /
fused-BN variants); everything else raises instead of silently computing something different.
"""
import math

import torch
from torch import nn

from .. import _lib
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


def _per_dim(value, ndim):
    return [int(v) for v in value] if isinstance(value, (list, tuple)) else [int(value)] * ndim


def _fans_hwio(weight):
    """(fan_in, fan_out) of a (k..., Cin, Cout) filter bank: receptive field x channels."""
    if weight.dim() < 2:
        raise ValueError('a filter bank needs at least (Cin, Cout)')
    field = 1
    for k in weight.shape[:-2]:
        field *= int(k)
    return weight.shape[-2] * field, weight.shape[-1] * field


class SparseConvolution(SparseModule):

    _UNSUPPORTED = ('transposed', 'inverse', 'fused_bn')

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        flags = dict(transposed=transposed, inverse=inverse, fused_bn=fused_bn)
        bad = [k for k in self._UNSUPPORTED if flags[k]]
        if bad or groups != 1:
            raise NotImplementedError('sparse conv variant off the DetMatch path: %s'
                                      % (bad or 'groups=%d' % groups))
        self.ndim, self.groups = ndim, groups
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _per_dim(kernel_size, ndim)
        self.stride = _per_dim(stride, ndim)
        self.padding = _per_dim(padding, ndim)
        self.dilation = _per_dim(dilation, ndim)
        self.output_padding = _per_dim(output_padding, ndim)
        if any(d != 1 and s != 1 for d, s in zip(self.dilation, self.stride)):
            raise ValueError('dilation and stride cannot both differ from 1')
        self.subm, self.indice_key = subm, indice_key
        self.transposed, self.inverse, self.fused_bn = transposed, inverse, fused_bn
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        """As the reference (conv.py:103-108): torch's kaiming_uniform_(a = sqrt(5)) applied to the
        (k..., Cin, Cout) tensor AS IS — torch derives its fan from dim 1, not from Cin, and that
        (smaller) scale is what released models were initialised with — and bias ~ U(+-1/sqrt(fan_in))
        with the true fan of the channels-last layout."""
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(_fans_hwio(self.weight)[0])
            nn.init.uniform_(self.bias, -bound, bound)

    # ------------------------------------------------------------------ forward
    def _rulebook(self, x):
        """(outids, indice_pairs, indice_pair_num) of this layer, shared through the tensor's
        `indice_dict` by every layer with the same `indice_key`."""
        hit = x.find_indice_pair(self.indice_key)
        if hit is not None:
            return hit[0], hit[2], hit[3]
        outids, pairs, num = ops.get_indice_pairs(
            x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride, self.padding,
            self.dilation, self.output_padding, self.subm, self.transposed, grid=x.grid)
        x.indice_dict[self.indice_key] = (outids, x.indices, pairs, num, x.spatial_shape)
        return outids, pairs, num

    def forward(self, input):
        if not isinstance(input, SparseConvTensor):
            raise TypeError('SparseConvolution expects a SparseConvTensor')
        if self.conv1x1:       # a pointwise layer touches no neighbours: one dense GEMM on the rows
            rows = _lib.blas_linear(input.features, self.weight.view(self.in_channels, self.out_channels).t())
            ids, shape = input.indices, input.spatial_shape
        else:
            ids, pairs, num = self._rulebook(input)
            conv = Fsp.indice_subm_conv if self.subm else Fsp.indice_conv
            rows = conv(input.features, self.weight, pairs, num, ids.shape[0])
            shape = input.spatial_shape if self.subm else ops.get_conv_output_size(
                input.spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)
        if self.bias is not None:
            rows = rows + self.bias
        out = SparseConvTensor(rows, ids, shape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


class SparseConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key)


class SubMConv3d(SparseConvolution):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, subm=True, indice_key=indice_key)
