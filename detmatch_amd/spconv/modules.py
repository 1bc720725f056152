"""`SparseModule` / `SparseSequential` with the reference's surface (mmdet3d/ops/spconv/modules.py:
30-137: positional / OrderedDict / keyword construction, integer indexing, `add`, `sparity_dict`,
child names "0", "1", ... so state-dict keys match) and this repo's execution: the container walks
a pre-resolved step list in which a BatchNorm1d followed by a ReLU is ONE fused row kernel
(csrc/bn_relu.hip) over the (N, C) feature rows.
"""
from collections import OrderedDict

from torch import nn

from .structure import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base: a SparseSequential hands these the SparseConvTensor itself, not its rows."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):

    def __init__(self, *args, **kwargs):
        super().__init__()
        self._sparity_dict = {}
        named = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            ((str(i), m) for i, m in enumerate(args))
        for name, module in list(named) + list(kwargs.items()):
            self._register_unique(name, module)

    def _register_unique(self, name, module):
        if name in self._modules:
            raise ValueError('module name %r is already taken' % name)
        self.add_module(name, module)

    def add(self, module, name=None):
        self._register_unique(str(len(self._modules)) if name is None else name, module)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        n = len(self._modules)
        if not -n <= idx < n:
            raise IndexError('index %d is out of range for %d children' % (idx, n))
        return list(self._modules.values())[idx % n]

    @property
    def sparity_dict(self):
        return self._sparity_dict

    # ------------------------------------------------------------------ execution
    def _steps(self):
        """[(kind, name, module, fused_relu)] with kind in {'sparse', 'bn', 'dense'}; a ReLU that
        directly follows a BatchNorm1d is folded into the 'bn' step."""
        children = list(self._modules.items())
        steps, i = [], 0
        while i < len(children):
            name, m = children[i]
            if is_spconv_module(m):
                steps.append(('sparse', name, m, False))
            elif isinstance(m, nn.BatchNorm1d):
                fused = i + 1 < len(children) and isinstance(children[i + 1][1], nn.ReLU)
                steps.append(('bn', name, m, fused))
                i += int(fused)
            else:
                steps.append(('dense', name, m, False))
            i += 1
        return steps

    def forward(self, input):
        x = input
        for kind, name, module, fused_relu in self._steps():
            sparse_in = isinstance(x, SparseConvTensor)
            if kind == 'sparse':
                if not sparse_in:
                    raise TypeError('%s expects a SparseConvTensor' % type(module).__name__)
                self._sparity_dict[name] = x.sparity
                x = module(x)
            elif not sparse_in:
                x = module(x)
                if fused_relu:
                    x = nn.functional.relu(x)
            elif x.indices.shape[0] != 0:         # dense layers act on the feature rows
                if kind == 'bn' and x.features.is_cuda:
                    from ..bn_relu import bn_relu_rows
                    x.features = bn_relu_rows(x.features, module, relu=fused_relu)
                else:
                    x.features = module(x.features)
                    if fused_relu:
                        x.features = nn.functional.relu(x.features)
        return x
