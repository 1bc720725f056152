"""mmdet3d/ops/spconv/modules.py:30-137 — SparseModule / SparseSequential."""
from collections import OrderedDict

from torch import nn

from .structure import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base: subclasses receive a SparseConvTensor inside SparseSequential."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Sequential container; dense modules are applied to `.features` (modules.py:125-137)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError('name exists')
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.items())
        skip = False
        for i, (k, module) in enumerate(mods):
            if skip:            # the ReLU fused into the preceding BatchNorm
                skip = False
                continue
            if isinstance(module, nn.BatchNorm1d) and isinstance(input, SparseConvTensor) and \
                    input.features.is_cuda and input.indices.shape[0] != 0:
                # BatchNorm1d (+ ReLU) over the (N, C) feature rows: one fused pass
                # (modules.py:125-137 runs them as two dense modules on input.features)
                from ..bn_relu import bn_relu_rows
                nxt = mods[i + 1][1] if i + 1 < len(mods) else None
                fuse = isinstance(nxt, nn.ReLU)
                input.features = bn_relu_rows(input.features, module, relu=fuse)
                skip = fuse
                continue
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = module(input.features)
            else:
                input = module(input)
        return input
