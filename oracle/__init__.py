"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of oracle/libdm_oracle.so (the plain-C CPU restatement of
the reference algorithms, see dm_oracle.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; detmatch_amd/ never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libdm_oracle.so')
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_intp = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C restatement (gcc, a second or two)."""
    src = os.path.join(_HERE, 'dm_oracle.c')
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'clean', 'all'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_box_overlap.restype = ctypes.c_float
        _lib.orc_iou_bev.restype = ctypes.c_float
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def _ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


# ---------------------------------------------------------------- voxelize
def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """-> voxels (V,max_points,C) f32, coors (V,3) i32 [z,y,x], num_points (V) i32."""
    points = _f32(points)
    n, c = points.shape
    voxels = np.zeros((max_voxels, max_points, c), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    vs = _f32(voxel_size)
    cr = _f32(coors_range)
    v = lib().orc_hard_voxelize(_p(points, _f32p), n, c, _p(vs, _f32p), _p(cr, _f32p),
                                int(max_points), int(max_voxels), _p(voxels, _f32p),
                                _p(coors, _i32p), _p(num, _i32p))
    return voxels[:v], coors[:v], num[:v]


# ---------------------------------------------------------------- spconv
def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """mmdet3d/ops/spconv/ops.py:20-31"""
    return [(input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1)
            // stride[i] + 1 for i in range(len(input_size))]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding,
                     dilation=(1, 1, 1), subm=False, sort_out=True):
    """-> outids (N_out,4), indice_pairs (K,2,N), indice_num (K), out_shape."""
    indices = _i32(indices)
    n = indices.shape[0]
    kvol = int(np.prod(ksize))
    if subm:
        out_shape = list(spatial_shape)
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    out_ids = np.zeros((max(kvol * n, 1), 4), np.int32)
    pairs = np.empty((kvol, 2, max(n, 1)), np.int32)
    num = np.zeros((kvol,), np.int32)
    if n == 0:
        return out_ids[:0], np.full((kvol, 2, 0), -1, np.int32), num, out_shape
    n_out = lib().orc_get_indice_pairs(
        _p(indices, _i32p), n, int(batch_size), _ints(out_shape), _ints(spatial_shape),
        _ints(ksize), _ints(stride), _ints(padding), _ints(dilation), int(subm),
        int(sort_out), _p(out_ids, _i32p), _p(pairs, _i32p), _p(num, _i32p))
    if n_out < 0:
        raise ValueError('batch_size * output volume overflows int32')
    return out_ids[:n_out].copy(), pairs, num, out_shape


def indice_conv(features, filters, indice_pairs, indice_num, n_out, subm=False):
    features = _f32(features)
    filters = _f32(filters)
    cin, cout = filters.shape[-2], filters.shape[-1]
    kvol = indice_pairs.shape[0]
    pairs = _i32(indice_pairs)
    num = _i32(indice_num)
    out = np.empty((n_out, cout), np.float32)
    lib().orc_indice_conv(_p(features, _f32p), features.shape[0], _p(filters, _f32p),
                          _p(pairs, _i32p), _p(num, _i32p), pairs.shape[2], kvol, cin, cout,
                          int(n_out), int(subm), _p(out, _f32p))
    return out


def indice_conv_backward(features, filters, out_grad, indice_pairs, indice_num, subm=False):
    features = _f32(features)
    filters = _f32(filters)
    out_grad = _f32(out_grad)
    cin, cout = filters.shape[-2], filters.shape[-1]
    kvol = indice_pairs.shape[0]
    pairs = _i32(indice_pairs)
    num = _i32(indice_num)
    in_grad = np.empty_like(features)
    filt_grad = np.empty_like(filters)
    lib().orc_indice_conv_backward(
        _p(features, _f32p), features.shape[0], _p(filters, _f32p), _p(out_grad, _f32p),
        out_grad.shape[0], _p(pairs, _i32p), _p(num, _i32p), pairs.shape[2], kvol, cin, cout,
        int(subm), _p(in_grad, _f32p), _p(filt_grad, _f32p))
    return in_grad, filt_grad


# ---------------------------------------------------------------- iou3d / nms
class host_libm_trig(object):
    """Context manager: evaluate cos/sin/atan2 of the box code with the FLOAT libm functions, as
    the compiled reference (g++: cos(float) -> cosf) does; default = correctly rounded."""

    def __enter__(self):
        lib().orc_set_trig_mode(1)

    def __exit__(self, *a):
        lib().orc_set_trig_mode(0)


def boxes_overlap_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_overlap_bev(_p(a, _f32p), a.shape[0], _p(b, _f32p), b.shape[0],
                                _p(out, _f32p))
    return out


def boxes_iou_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_iou_bev(_p(a, _f32p), a.shape[0], _p(b, _f32p), b.shape[0],
                            _p(out, _f32p))
    return out


def boxes_iou3d(a, b):
    """pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-81 (height overlap in fp32)."""
    a, b = _f32(a), _f32(b)
    overlaps_bev = boxes_overlap_bev(a, b)
    a_max = (a[:, 2] + a[:, 5] / np.float32(2)).reshape(-1, 1)
    a_min = (a[:, 2] - a[:, 5] / np.float32(2)).reshape(-1, 1)
    b_max = (b[:, 2] + b[:, 5] / np.float32(2)).reshape(1, -1)
    b_min = (b[:, 2] - b[:, 5] / np.float32(2)).reshape(1, -1)
    overlaps_h = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (a[:, 3] * a[:, 4] * a[:, 5]).reshape(-1, 1)
    vol_b = (b[:, 3] * b[:, 4] * b[:, 5]).reshape(1, -1)
    return (overlaps_3d / np.clip(vol_a + vol_b - overlaps_3d, np.float32(1e-6), None)
            ).astype(np.float32)


def nms(boxes_sorted, thresh, normal=False):
    """boxes already sorted by descending score -> kept positions (int64)."""
    boxes = _f32(boxes_sorted)
    keep = np.zeros((boxes.shape[0],), np.int64)
    fn = lib().orc_nms_normal if normal else lib().orc_nms
    k = fn(_p(boxes, _f32p), boxes.shape[0], ctypes.c_float(thresh), _p(keep, _i64p))
    return keep[:k]


# ---------------------------------------------------------------- pointnet2 stack
def ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """pointnet2_utils.py:8-38 incl. the empty-ball post-processing.
    -> idx (M,nsample) i32, empty_ball_mask (M) bool."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), np.int32)
    lib().orc_ball_query_stack(len(xc), m, ctypes.c_float(radius), int(nsample),
                               _p(new_xyz, _f32p), _p(nc, _i32p), _p(xyz, _f32p),
                               _p(xc, _i32p), _p(idx, _i32p))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def group_points(features, features_batch_cnt, idx, idx_batch_cnt):
    features, idx = _f32(features), _i32(idx)
    fc, ic = _i32(features_batch_cnt), _i32(idx_batch_cnt)
    m, ns = idx.shape
    c = features.shape[1]
    out = np.empty((m, c, ns), np.float32)
    lib().orc_group_points_stack(len(fc), m, c, ns, _p(features, _f32p), _p(fc, _i32p),
                                 _p(idx, _i32p), _p(ic, _i32p), _p(out, _f32p))
    return out


def group_points_grad(grad_out, idx, idx_batch_cnt, features_batch_cnt, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    fc, ic = _i32(features_batch_cnt), _i32(idx_batch_cnt)
    m, c, ns = grad_out.shape
    out = np.empty((n, c), np.float32)
    lib().orc_group_points_grad_stack(len(fc), m, c, int(n), ns, _p(grad_out, _f32p),
                                      _p(idx, _i32p), _p(ic, _i32p), _p(fc, _i32p),
                                      _p(out, _f32p))
    return out


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) -> idx (B,npoint) i32 (pointnet2_utils.py:158-180)."""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    out = np.zeros((b, npoint), np.int32)
    lib().orc_furthest_point_sampling(b, n, int(npoint), _p(xyz, _f32p), _p(temp, _f32p),
                                      _p(out, _i32p))
    return out


def points_in_boxes(points, boxes):
    """points (B,P,3), boxes (B,T,7) -> (B,P) i32 (roiaware_pool3d_utils.py:28-41)."""
    points, boxes = _f32(points), _f32(boxes)
    b, p, _ = points.shape
    out = np.empty((b, p), np.int32)
    lib().orc_points_in_boxes(b, boxes.shape[1], p, _p(boxes, _f32p), _p(points, _f32p),
                              _p(out, _i32p))
    return out


def roi_align(feat, rois, spatial_scale, out_size=7, sampling_ratio=0, aligned=True):
    """feat (N,C,H,W), rois (R,5) -> (R,C,out,out); mmcv RoIAlign avg (parity unpinned)."""
    feat, rois = _f32(feat), _f32(rois)
    n, c, h, w = feat.shape
    r = rois.shape[0]
    out = np.zeros((r, c, out_size, out_size), np.float32)
    lib().orc_roi_align_forward(_p(feat, _f32p), c, h, w, _p(rois, _f32p), r,
                                ctypes.c_float(spatial_scale), out_size, out_size, sampling_ratio,
                                int(aligned), _p(out, _f32p))
    return out


def roi_align_grad(grad_out, feat_shape, rois, spatial_scale, sampling_ratio=0, aligned=True):
    """-> d feat (N,C,H,W) float64"""
    grad_out, rois = _f32(grad_out), _f32(rois)
    n, c, h, w = feat_shape
    r, _, ph, pw = grad_out.shape
    g = np.zeros(feat_shape, np.float64)
    lib().orc_roi_align_backward(_p(grad_out, _f32p), c, h, w, _p(rois, _f32p), r,
                                 ctypes.c_float(spatial_scale), ph, pw, sampling_ratio, int(aligned),
                                 g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return g


def points_augment(points, params, perm=None):
    """One view of dm_points_augment: points (N,C) f32, params (24,) f32 -> kept rows (M,C)."""
    points, params = _f32(points), _f32(params)
    n, c = points.shape
    out = np.zeros((n, c), np.float32)
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        pp = perm.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    L = lib()
    L.orc_points_augment.restype = ctypes.c_int
    kept = L.orc_points_augment(_p(points, _f32p), n, c, _p(params, _f32p), pp, _p(out, _f32p))
    return out[:kept]
