"""BaseBEVBackbone — pcdet/models/backbones_2d/base_bev_backbone.py:9-117 (dense 2-D convs;
MIOpen / hipBLASLt through torch — the legitimately MFMA-bound part of the 3D branch)."""
from functools import partial

import numpy as np
import torch
import torch.nn as nn


class BaseBEVBackbone(nn.Module):

    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm2d, eps=1e-3, momentum=0.01)
        layer_nums = model_cfg.get('LAYER_NUMS', None) or []
        layer_strides = model_cfg.get('LAYER_STRIDES', None) or []
        num_filters = model_cfg.get('NUM_FILTERS', None) or []
        upsample_strides = model_cfg.get('UPSAMPLE_STRIDES', None) or []
        num_upsample_filters = model_cfg.get('NUM_UPSAMPLE_FILTERS', None) or []
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        assert len(upsample_strides) == len(num_upsample_filters)
        num_levels = len(layer_nums)
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(num_levels):
            cur_layers = [nn.ZeroPad2d(1),
                          nn.Conv2d(c_in_list[idx], num_filters[idx], kernel_size=3,
                                    stride=layer_strides[idx], padding=0, bias=False),
                          norm_fn(num_filters[idx]), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                cur_layers.extend([nn.Conv2d(num_filters[idx], num_filters[idx], kernel_size=3,
                                             padding=1, bias=False),
                                   norm_fn(num_filters[idx]), nn.ReLU()])
            self.blocks.append(nn.Sequential(*cur_layers))
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    self.deblocks.append(nn.Sequential(
                        nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx],
                                           upsample_strides[idx], stride=upsample_strides[idx],
                                           bias=False),
                        norm_fn(num_upsample_filters[idx]), nn.ReLU()))
                else:
                    stride = int(np.round(1 / stride))
                    self.deblocks.append(nn.Sequential(
                        nn.Conv2d(num_filters[idx], num_upsample_filters[idx], stride,
                                  stride=stride, bias=False),
                        norm_fn(num_upsample_filters[idx]), nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > num_levels:
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1],
                                   bias=False), norm_fn(c_in), nn.ReLU()))
        self.num_bev_features = c_in

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups = []
        x = spatial_features
        for i in range(len(self.blocks)):
            x = self.blocks[i](x)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            ups.append(self.deblocks[i](x) if len(self.deblocks) > 0 else x)
        if len(ups) > 1:
            x = torch.cat(ups, dim=1)
        elif len(ups) == 1:
            x = ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        data_dict['spatial_features_2d'] = x
        return data_dict
