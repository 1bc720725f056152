"""ResNet-50 (caffe style) + FPN, mmdet 2.14 layouts and parameter names (parity unpinned).

Config: depth 50, out_indices (0,1,2,3), frozen_stages 1, norm_cfg BN requires_grad False,
norm_eval True, style 'caffe' (the stride-2 sits in the first 1x1 conv of a bottleneck).

With norm_eval + requires_grad False every BatchNorm of the backbone is a constant per-channel
affine map, so `FrozenBNConv` keeps the reference parameters (conv.weight, bn.weight, bn.bias,
bn.running_mean, bn.running_var — same state-dict keys) but runs conv(x, w * s) + b with
s = gamma / sqrt(var + eps), b = beta - mean * s: the BN kernel and its HBM round trip disappear
while gradients w.r.t. conv.weight stay exact (d/dw of conv(x, w*s) = s * d/dw' ).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dense_conv
from ..fused import on as fused_on


class FrozenBN(nn.Module):
    """State-dict compatible with nn.BatchNorm2d (weight, bias, running_mean, running_var,
    num_batches_tracked); never updates."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    # bumped by whoever rewrites these tensors behind autograd's back (raw-pointer kernels such as the
    # fused EMA); ordinary in-place updates (load_state_dict, copy_) are caught by Tensor._version
    GENERATION = 0

    def scale_shift(self):
        """The constant affine map of the frozen layer: (gamma / sqrt(var + eps), beta - mean * scale),
        cached until one of the four tensors changes."""
        key = (FrozenBN.GENERATION, self.weight._version, self.bias._version,
               self.running_mean._version, self.running_var._version, self.weight.device)
        cached = getattr(self, '_affine', None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                s = self.weight * torch.rsqrt(self.running_var + self.eps)
                b = self.bias - self.running_mean * s
            s.dm_constant = True      # lives as long as it is valid: packed weights may keep its address
            cached = self._affine = (key, s, b)
        return cached[1], cached[2]


def conv_frozen_bn(x, conv, bn, relu, residual=None):
    """relu?(conv(x, w * s) + b [+ residual]): the fold happens while the weights are packed for the
    kernel (`w_scale`), bias, shortcut and ReLU in the GEMM's epilogue; gradients come back w.r.t. the
    unscaled conv.weight (csrc/conv2d.hip, dense_conv.conv2d)."""
    s, b = bn.scale_shift()
    return dense_conv.conv2d(x, conv.weight, b, conv.stride, conv.padding, relu=relu, w_scale=s,
                             residual=residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        # caffe style: stride on conv1
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = FrozenBN(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn2 = FrozenBN(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBN(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            FrozenBN(planes * 4))

    def forward(self, x):
        out = conv_frozen_bn(x, self.conv1, self.bn1, True)
        out = conv_frozen_bn(out, self.conv2, self.bn2, True)
        idt = x if self.downsample is None else conv_frozen_bn(x, self.downsample[0],
                                                               self.downsample[1], False)
        if fused_on() and x.is_cuda:
            # relu(bn3(conv3(out)) + identity) in conv3's epilogue: the sum and the ReLU were two more
            # passes over the block's output (the largest tensor of the block)
            return conv_frozen_bn(out, self.conv3, self.bn3, True, residual=idt)
        out = conv_frozen_bn(out, self.conv3, self.bn3, False)
        return F.relu_(out + idt)


class ResNet(nn.Module):
    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=None, norm_eval=True, style='pytorch', init_cfg=None,
                 zero_init_residual=True, **kwargs):
        super().__init__()
        assert style == 'caffe' and norm_eval and norm_cfg is not None and \
            not norm_cfg.get('requires_grad', True), \
            'only the DetMatch configuration (caffe style, frozen eval-mode BN) is built'
        self.out_indices = out_indices
        self.frozen_stages = frozen_stages
        self.zero_init_residual = zero_init_residual
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBN(64)
        inplanes = 64
        self.res_layers = []
        for i, nblocks in enumerate(self.arch[depth][:num_stages]):
            planes, stride = 64 * 2 ** i, (1 if i == 0 else 2)
            blocks = [Bottleneck(inplanes, planes, stride, downsample=True)]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, nblocks)]
            name = 'layer%d' % (i + 1)
            setattr(self, name, nn.Sequential(*blocks))
            self.res_layers.append(name)
        self._freeze_stages()
        self.init_weights()

    def init_weights(self):
        """mmdet ResNet.init_weights without a checkpoint: kaiming convs, unit BN, and
        zero_init_residual (default True) zeroes the last BN scale of every bottleneck."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.conv1.weight.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, 'layer%d' % i).parameters():
                p.requires_grad = False

    def forward(self, x):
        x = conv_frozen_bn(x, self.conv1, self.bn1, True)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class ConvModule(nn.Module):
    """mmcv ConvModule without norm/activation: holds `.conv` so keys read `x.conv.weight`."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = dense_conv.Conv2d(cin, cout, k, padding=padding)
        nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    """mmdet FPN (add_extra_convs False): laterals 1x1, top-down nearest x2 upsampling adds,
    3x3 output convs, extra levels by max_pool2d(1, stride=2)."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, **kwargs):
        super().__init__()
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.lateral_convs = nn.ModuleList([ConvModule(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([ConvModule(out_channels, out_channels, 3, padding=1)
                                        for _ in in_channels])

    def forward(self, inputs):
        if fused_on() and inputs[0].is_cuda:
            # top-down: lateral(x_i) + upsample(lat_{i+1}) with the sum in the lateral conv's epilogue
            lat = [None] * len(inputs)
            lat[-1] = self.lateral_convs[-1](inputs[-1])
            for i in range(len(inputs) - 2, -1, -1):
                c = self.lateral_convs[i].conv
                up = F.interpolate(lat[i + 1], size=inputs[i].shape[2:], mode='nearest')
                lat[i] = dense_conv.conv2d(inputs[i], c.weight, c.bias, c.stride, c.padding, residual=up)
        else:
            lat = [l(x) for l, x in zip(self.lateral_convs, inputs)]
            for i in range(len(lat) - 1, 0, -1):
                lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [c(x) for c, x in zip(self.fpn_convs, lat)]
        for _ in range(self.num_outs - len(outs)):
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)
