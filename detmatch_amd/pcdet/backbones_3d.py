"""MeanVFE, VoxelBackBone8x, HeightCompression.

Reference: pcdet/models/backbones_3d/vfe/mean_vfe.py:14-28,
pcdet/models/backbones_3d/spconv_backbone.py:9-28,68-175,
pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-25.
"""
from functools import partial

import torch
import torch.nn as nn

from .. import spconv


class MeanVFE(nn.Module):
    """mean of the points in a voxel.  When the batch dict already carries
    `voxel_features` (the fused mean written by dm_hard_voxelize) it is kept."""

    def __init__(self, model_cfg=None, num_point_features=4, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        if batch_dict.get('voxel_features', None) is not None:
            return batch_dict
        voxel_features, voxel_num_points = batch_dict['voxels'], batch_dict['voxel_num_points']
        points_mean = voxel_features.sum(dim=1, keepdim=False)
        normalizer = torch.clamp_min(voxel_num_points.view(-1, 1), min=1.0).type_as(voxel_features)
        batch_dict['voxel_features'] = (points_mean / normalizer).contiguous()
        return batch_dict


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type='subm', norm_fn=None):
    """spconv_backbone.py:9-28: conv -> BatchNorm1d -> ReLU."""
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False,
                                 indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride,
                                   padding=padding, bias=False, indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class VoxelBackBone8x(nn.Module):
    """12 sparse convs / 8 rulebooks (spconv_backbone.py:80-120)."""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg if model_cfg is not None else {}
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.frozen_stages = self.model_cfg.get('FROZEN_STAGES', -1)
        gs = [int(v) for v in grid_size]
        self.sparse_shape = [gs[2] + 1, gs[1], gs[0]]  # grid_size[::-1] + [1, 0, 0]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'),
            norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(
            block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2',
                  conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3',
                  conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4',
                  conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = self.model_cfg.get('last_pad', 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.conv_input.eval()
            for p in self.conv_input.parameters():
                p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, 'conv%d' % i)
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    @torch.no_grad()
    def build_rulebooks(self, voxel_coords, batch_size):
        """The 8 rulebooks of the 12 convolutions, from the voxel coordinates alone (they do not
        depend on the weights): {indice_key: (outids, in_indices, indice_pairs, indice_num, shape)},
        the cache format of SparseConvolution.forward (spconv/conv.py:146-172)."""
        from ..spconv import ops as sp_ops
        return sp_ops.drive_steps(self.build_rulebooks_steps(voxel_coords, batch_size))

    def build_rulebooks_steps(self, voxel_coords, batch_size, ws_tag='rulebook'):
        """build_rulebooks as a generator that yields the four N_out device scalars it needs
        (spconv/ops.py:drive_steps_together reads those of several passes back in one copy)."""
        from ..spconv import ops as sp_ops
        from ..spconv.conv import SparseConvolution
        indice_dict = {}
        indices, shape = voxel_coords.int(), list(self.sparse_shape)
        for m in self.modules():
            if not isinstance(m, SparseConvolution) or m.conv1x1:
                continue
            out_shape = shape if m.subm else sp_ops.get_conv_output_size(
                shape, m.kernel_size, m.stride, m.padding, m.dilation)
            if m.indice_key not in indice_dict:
                outids, pairs, num = yield from sp_ops.get_indice_pairs_steps(
                    indices, batch_size, shape, m.kernel_size, m.stride, m.padding, m.dilation,
                    m.output_padding, m.subm, m.transposed, ws_tag=ws_tag)
                indice_dict[m.indice_key] = (outids, indices, pairs, num, shape)
            indices, shape = indice_dict[m.indice_key][0], out_shape
        return indice_dict

    def build_rulebooks_deferred(self, coords_cap, n_dev, batch_size, ws_tag='rulebook'):
        """Every rulebook of the backbone issued WITHOUT a host value: `coords_cap` (cap, 4) voxel coordinates at the
        voxelizer's capacity, `n_dev` its device count.  -> (levels, counts): levels = [(indice_key, CapRulebook,
        input spatial shape)], counts = the device scalars N_out of the strided levels in order — the caller reads
        them (with the voxel count) in ONE copy and calls finish_rulebooks.  None: the capacity-sized entries do not
        apply (tiny clouds on huge grids: the occupancy bitmap does not fit) — use build_rulebooks_steps."""
        from ..spconv import ops as sp_ops
        from ..spconv.conv import SparseConvolution
        levels, counts, seen = [], [], {}
        indices, n_cur, cap_cur, shape = coords_cap, n_dev, int(coords_cap.shape[0]), list(self.sparse_shape)
        first_cap = cap_cur
        for m in self.modules():
            if not isinstance(m, SparseConvolution) or m.conv1x1:
                continue
            if m.transposed:
                return None
            out_shape = shape if m.subm else sp_ops.get_conv_output_size(
                shape, m.kernel_size, m.stride, m.padding, m.dilation)
            if m.indice_key not in seen:
                cap_out = None if m.subm else sp_ops.strided_capacity(cap_cur, m.kernel_size, m.stride, first_cap)
                c = sp_ops.build_rulebook_cap(indices, n_cur, cap_cur, batch_size, shape, m.kernel_size, m.stride,
                                              m.padding, m.dilation, m.subm, cap_out=cap_out, ws_tag=ws_tag)
                if c is None:
                    return None
                seen[m.indice_key] = c
                levels.append((m.indice_key, c, shape))
                if not m.subm:
                    counts.append(c.n_out_dev)
            c = seen[m.indice_key]
            indices, n_cur, cap_cur, shape = c.outids, c.n_out_dev, c.cap_out, out_shape
        return levels, counts

    @staticmethod
    def finish_rulebooks(levels, n_voxels, n_outs):
        """The host counts are here: exact-size rulebooks -> the indice_dict build_rulebooks_steps returns, or None when
        a level outgrew its capacity."""
        indice_dict, n_in, k = {}, int(n_voxels), 0
        for key, c, shape in levels:
            if c.subm:
                n_out = n_in
            else:
                n_out = int(n_outs[k])
                k += 1
            rb = c.finish(n_in, n_out)
            if rb is None:
                return None
            rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
            indice_dict[key] = (rb.outids, c.indices[:n_in], rb.indice_pairs, rb.indice_num, shape)
            n_in = n_out
        return indice_dict

    def _chained(self, batch_dict, voxel_features, voxel_coords):
        """The 12 layers as one chained call (sparse_chain.py), when it applies."""
        from .. import chain as _chain
        if not _chain.on('sparse') or not voxel_features.is_cuda or voxel_features.dtype != torch.float32 \
                or voxel_features.shape[0] < 2:
            return False
        from ..sparse_chain import SparseBackboneChain
        train = self.conv_input[1].training
        if torch.is_grad_enabled() and (voxel_features.requires_grad or
                                        (not train and any(p.requires_grad for p in self.parameters()))):
            # the chain returns no gradient for the voxel features (MeanVFE has no parameters) and, in evaluation
            # mode, none at all: whoever needs one takes the op-by-op path
            return False
        cache = self.__dict__.setdefault('_chains', {})
        ch = cache.get(train)
        if ch is None or (ch is not False and not ch.valid()):
            ch = cache[train] = SparseBackboneChain(self, voxel_features.device, train) \
                if SparseBackboneChain.applicable(self, train) else False
        if ch is False:
            return False
        indice_dict = dict(batch_dict.pop('indice_dict_prefetch', None) or {})
        x1, x2, x3, x4, out = ch(voxel_features, voxel_coords.int(), batch_dict['batch_size'], self.sparse_shape, indice_dict)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv1': x1, 'x_conv2': x2, 'x_conv3': x3, 'x_conv4': x4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return True

    def forward(self, batch_dict):
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        if self._chained(batch_dict, voxel_features, voxel_coords):
            return batch_dict
        x = spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords.int(),
                                    spatial_shape=self.sparse_shape,
                                    batch_size=batch_dict['batch_size'])
        x.indice_dict.update(batch_dict.pop('indice_dict_prefetch', None) or {})
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {
            'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4}})
        batch_dict.update({'multi_scale_3d_strides': {
            'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class _HeightCompressFn(torch.autograd.Function):
    """dm_height_compress_forward / _backward (csrc/chain_ops.hip)."""

    @staticmethod
    def forward(ctx, features, indices, batch, d, h, w):
        from .. import _lib
        features = features.contiguous()
        n, c = features.shape
        out = torch.empty((batch, h, w, c * d), dtype=torch.float32, device=features.device)
        _lib.check(_lib.lib().dm_height_compress_forward(features.data_ptr(), indices.data_ptr(), n, c, batch, d, h, w,
                                                         out.data_ptr(), _lib.raw_stream()), 'dm_height_compress_forward')
        ctx.save_for_backward(indices)
        ctx.geom = (n, c, d, h, w)
        return out

    @staticmethod
    def backward(ctx, grad):
        from .. import _lib
        (indices,) = ctx.saved_tensors
        n, c, d, h, w = ctx.geom
        grad = grad.contiguous()
        g = torch.empty((n, c), dtype=torch.float32, device=grad.device)
        _lib.check(_lib.lib().dm_height_compress_backward(grad.data_ptr(), indices.data_ptr(), n, c, d, h, w, g.data_ptr(),
                                                          _lib.raw_stream()), 'dm_height_compress_backward')
        return g, None, None, None, None, None


class HeightCompression(nn.Module):
    """(B, C, D, H, W) dense scatter viewed as (B, C*D, H, W)."""

    def __init__(self, model_cfg=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg if model_cfg is not None else {}
        self.num_bev_features = self.model_cfg.get('NUM_BEV_FEATURES', 256)

    def forward(self, batch_dict):
        """height_compression.py:10-26: `.dense()` -> (B, C, D, H, W) -> view (B, C*D, H, W), i.e.
        channel c*D + d.  Built here directly in NHWC memory (what the BEV convolutions read): the
        active voxels are scattered into a zeroed (B, H, W, C, D) buffer, whose (B, H, W, C*D) view
        permuted to (B, C*D, H, W) is the reference tensor in torch.channels_last layout."""
        sp = batch_dict['encoded_spconv_tensor']
        d, h, w = (int(v) for v in sp.spatial_shape)
        c = sp.features.shape[1]
        from ..fused import on as fused_on
        if fused_on() and sp.features.is_cuda and sp.features.dtype == torch.float32 and sp.indices.dtype == torch.int32 \
                and sp.indices.is_contiguous() and sp.indices.shape[1] == 4:
            buf = _HeightCompressFn.apply(sp.features, sp.indices, int(sp.batch_size), d, h, w)
            batch_dict['spatial_features'] = buf.permute(0, 3, 1, 2)
            batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
            return batch_dict
        idx = sp.indices.long()
        buf = sp.features.new_zeros((sp.batch_size, h, w, c, d))
        buf[idx[:, 0], idx[:, 2], idx[:, 3], :, idx[:, 1]] = sp.features      # (N, C) rows
        batch_dict['spatial_features'] = buf.view(sp.batch_size, h, w, c * d).permute(0, 3, 1, 2)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
