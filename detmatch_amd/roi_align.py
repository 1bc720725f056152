"""RoIAlign over an FPN pyramid — host mirror of mmcv.ops.RoIAlign / mmdet SingleRoIExtractor
(mmcv-full 1.3.16 / mmdet 2.14.0, un-vendored: parity unpinned) as used at
configs/detmatch/001/detmatch/split_0.py:74-80 and processors_2d.py:52-54.

One dm_roi_align_forward launch serves all pyramid levels (each RoI carries its level);
mmdet runs one masked launch per level with a nonzero() host sync in between."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib


def _pyramid_args(tensors):
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    hs = (ctypes.c_int32 * n)(*[int(t.shape[2]) for t in tensors])
    ws = (ctypes.c_int32 * n)(*[int(t.shape[3]) for t in tensors])
    return ptrs, hs, ws


class _RoIAlignFPN(Function):

    @staticmethod
    def forward(ctx, rois, levels, scales, out_size, sampling_ratio, aligned, *feats):
        if not feats[0].is_cuda:
            _lib.require_device(feats[0])      # raises: no CPU path
        assert all(f.dtype == torch.float32 and f.shape[1] == feats[0].shape[1] for f in feats)
        rois = rois.contiguous().float()
        levels = levels.contiguous().int() if levels is not None else None
        r, c = rois.shape[0], feats[0].shape[1]
        out = torch.empty((r, c, out_size, out_size), dtype=torch.float32, device=feats[0].device)
        # channel-last maps (a no-op for the channels_last tensors the convolutions produce): every tap
        # then reads 256 contiguous bytes per wave
        nhwc = [f.permute(0, 2, 3, 1).contiguous() for f in feats]
        ptrs = (ctypes.c_void_p * len(nhwc))(*[t.data_ptr() for t in nhwc])
        hs = (ctypes.c_int32 * len(feats))(*[int(f.shape[2]) for f in feats])
        ws = (ctypes.c_int32 * len(feats))(*[int(f.shape[3]) for f in feats])
        sc = (ctypes.c_float * len(feats))(*[float(s) for s in scales])
        _lib.check(_lib.lib().dm_roi_align_forward_nhwc(
            ptrs, hs, ws, sc, len(feats), c, _lib.ptr(rois), _lib.ptr(levels) if levels is not None else None,
            r, out_size, out_size, sampling_ratio, int(aligned), _lib.ptr(out), _lib.stream()),
            'dm_roi_align_forward_nhwc')
        ctx.save_for_backward(rois, levels)
        ctx.meta = (scales, out_size, sampling_ratio, aligned, [tuple(f.shape) for f in feats])
        return out

    @staticmethod
    def backward(ctx, gout):
        rois, levels = ctx.saved_tensors
        scales, out_size, sampling_ratio, aligned, shapes = ctx.meta
        gout = gout.contiguous()
        # accumulate in NHWC (channel-contiguous float atomics), hand back NCHW-shaped views
        grads = [gout.new_zeros((s[0], s[2], s[3], s[1])) for s in shapes]
        ptrs = (ctypes.c_void_p * len(grads))(*[t.data_ptr() for t in grads])
        hs = (ctypes.c_int32 * len(grads))(*[int(s[2]) for s in shapes])
        ws = (ctypes.c_int32 * len(grads))(*[int(s[3]) for s in shapes])
        sc = (ctypes.c_float * len(grads))(*[float(s) for s in scales])
        _lib.check(_lib.lib().dm_roi_align_backward_nhwc(
            ptrs, hs, ws, sc, len(grads), shapes[0][1], _lib.ptr(rois),
            _lib.ptr(levels) if levels is not None else None, rois.shape[0], out_size, out_size,
            sampling_ratio, int(aligned), _lib.ptr(gout), _lib.stream()), 'dm_roi_align_backward_nhwc')
        grads = [t.permute(0, 3, 1, 2) for t in grads]
        return (None, None, None, None, None, None) + tuple(grads)


def roi_align(feat, rois, output_size=7, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """mmcv.ops.roi_align on one map: feat (N,C,H,W), rois (R,5) -> (R,C,out,out)."""
    return _RoIAlignFPN.apply(rois, None, [spatial_scale], output_size, sampling_ratio, aligned, feat)


def map_roi_levels(rois, num_levels, finest_scale=56):
    """mmdet SingleRoIExtractor.map_roi_levels: scale = sqrt(w*h);
    level = clamp(floor(log2(scale / finest_scale + 1e-6)), 0, num_levels-1)."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvl = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvl.clamp(min=0, max=num_levels - 1).int()


def roi_align_fpn(feats, rois, featmap_strides, output_size=7, sampling_ratio=0, aligned=True,
                  finest_scale=56):
    """SingleRoIExtractor.forward: feats list of (N,C,H_l,W_l), rois (R,5) -> (R,C,out,out)."""
    feats = list(feats)[:len(featmap_strides)]
    levels = map_roi_levels(rois, len(feats), finest_scale) if len(feats) > 1 else None
    return _RoIAlignFPN.apply(rois, levels, [1.0 / s for s in featmap_strides], output_size,
                              sampling_ratio, aligned, *feats)
