"""Generates tests/golden/ssl_geometry.npz by running the REFERENCE's own SSL-side geometry
(loaded by file path from /root/reference; build container only):

    mmdet3d/core/bbox/structures/{utils,base_box3d,lidar_box3d}.py   LiDARInstance3DBoxes
    mmdet3d/models/ssl_modules/bbox_utils.py                         apply_3d_transformation_bboxes,
                                                                     bbox_3d_to_bbox_2d
    mmdet3d/models/fusion_layers/coord_transform.py                  bbox_2d_transform

The reference package itself is not importable here (mmcv / mmdet / mmseg are absent), so the
files are loaded under their real dotted names after registering EMPTY placeholder modules
for the imports they do not use on this path.  The fixture holds inputs + reference outputs
only.

    python tests/golden/gen_ssl_geometry_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(dotted, rel):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    class BasePoints(object):
        pass
    for n in ('mmdet3d', 'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.core.bbox.structures',
              'mmdet3d.ops', 'mmdet3d.models', 'mmdet3d.models.fusion_layers',
              'mmdet3d.models.ssl_modules', 'mmcv', 'mmcv.ops', 'mmdet', 'mmdet.core',
              'mmdet.core.bbox'):
        _stub(n)
    _stub('mmdet3d.core.points', BasePoints=BasePoints, get_points_type=lambda *a, **k: None)
    _stub('mmdet3d.ops.roiaware_pool3d', points_in_boxes_gpu=None)
    _stub('mmdet3d.ops.iou3d', iou3d_cuda=None)
    _stub('mmcv.ops.nms', batched_nms=None)
    _stub('mmdet.core.bbox.iou_calculators', bbox_overlaps=None)
    sys.modules['mmdet3d.ops'].roiaware_pool3d = sys.modules['mmdet3d.ops.roiaware_pool3d']
    _load('mmdet3d.core.bbox.structures.utils', 'mmdet3d/core/bbox/structures/utils.py')
    _load('mmdet3d.core.bbox.structures.base_box3d', 'mmdet3d/core/bbox/structures/base_box3d.py')
    lb = _load('mmdet3d.core.bbox.structures.lidar_box3d',
               'mmdet3d/core/bbox/structures/lidar_box3d.py')
    core = sys.modules['mmdet3d.core']
    core.box3d_multiclass_nms = None
    core.xywhr2xyxyr = None
    core.bbox3d2result = None
    core.LiDARInstance3DBoxes = lb.LiDARInstance3DBoxes
    sys.modules['mmdet3d.core.bbox'].LiDARInstance3DBoxes = lb.LiDARInstance3DBoxes
    sys.modules['mmdet3d.core'].apply_3d_transformation = None
    bu = _load('mmdet3d.models.ssl_modules.bbox_utils', 'mmdet3d/models/ssl_modules/bbox_utils.py')
    ct = _load('mmdet3d.models.fusion_layers.coord_transform',
               'mmdet3d/models/fusion_layers/coord_transform.py')
    return lb.LiDARInstance3DBoxes, bu, ct


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from detmatch_amd import synth
    Boxes, bu, ct = load_reference()
    rng = np.random.default_rng(0)
    out = {}
    # boxes in front of the camera (KITTI-like), mm3d convention (x,y,z_bottom,w,l,h,yaw)
    n = 24
    b = np.stack([rng.uniform(4, 60, n), rng.uniform(-12, 12, n), rng.uniform(-2.0, -1.0, n),
                  rng.uniform(0.5, 2.0, n), rng.uniform(0.6, 4.5, n), rng.uniform(1.2, 2.0, n),
                  rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    b[0, 0] = -5.0      # behind the camera -> invalid
    b[1, 1] = 60.0      # far to the side  -> outside the image
    out['boxes'] = b
    boxes = Boxes(torch.from_numpy(b))
    out['corners'] = boxes.corners.numpy()
    out['gravity_center'] = boxes.gravity_center.numpy()
    out['from_gravity_origin'] = Boxes(torch.from_numpy(b), origin=(0.5, 0.5, 0.5)).tensor.numpy()
    lidar2img = synth.KITTI_LIDAR2IMG
    out['lidar2img'] = lidar2img
    xyxy, valid = bu.bbox_3d_to_bbox_2d(boxes, lidar2img, (375, 1242, 3))
    out['proj_xyxy'], out['proj_valid'] = xyxy.numpy(), valid.numpy()
    # augmentation replay: student meta ['HF','R','S','T'] and teacher meta ['HF']
    th = 0.31
    c, s = np.cos(th), np.sin(th)
    M = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], np.float32)    # unlabeled-sample form
    meta = dict(pcd_rotation=torch.from_numpy(M), pcd_scale_factor=1.04,
                pcd_trans=np.array([0.1, -0.2, 0.05], np.float32), pcd_horizontal_flip=True,
                pcd_vertical_flip=False, transformation_3d_flow=['HF', 'R', 'S', 'T'])
    out['meta_rotation'], out['meta_scale'], out['meta_trans'] = M, np.float32(1.04), meta['pcd_trans']
    fwd = bu.apply_3d_transformation_bboxes(boxes, meta, reverse=False)
    out['aug_forward'] = fwd.tensor.numpy()
    out['aug_roundtrip'] = bu.apply_3d_transformation_bboxes(fwd, meta, reverse=True).tensor.numpy()
    # 2-D transform
    b2 = np.stack([rng.uniform(0, 600, 16), rng.uniform(0, 180, 16), rng.uniform(620, 1240, 16),
                   rng.uniform(190, 370, 16), rng.uniform(0, 1, 16)], 1).astype(np.float32)
    meta2 = dict(img_shape=(384, 1272, 3), ori_shape=(375, 1242, 3),
                 scale_factor=np.array([1.0241546, 1.024, 1.0241546, 1.024], np.float32), flip=True)
    out['boxes2d'] = b2
    out['meta2_scale'] = meta2['scale_factor']
    o2n = ct.bbox_2d_transform(meta2, torch.from_numpy(b2), True)
    out['boxes2d_ori2new'] = o2n.numpy()
    out['boxes2d_back'] = ct.bbox_2d_transform(meta2, o2n, False).numpy()
    np.savez_compressed(os.path.join(HERE, 'ssl_geometry.npz'), **out)
    print('wrote ssl_geometry.npz; valid projections:', int(valid.sum()), 'of', n)


if __name__ == '__main__':
    main()
