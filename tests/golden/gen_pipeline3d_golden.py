"""Generates tests/golden/pipeline3d.npz by running the REFERENCE's own 3D augmentation transforms
(loaded by file path from /root/reference; build container only) on seeded frames:

    mmdet3d/core/points/{base_points,lidar_points}.py                 LiDARPoints
    mmdet3d/core/bbox/structures/{utils,base_box3d,lidar_box3d}.py    LiDARInstance3DBoxes
    mmdet3d/datasets/pipelines/transforms_3d.py    RandomFlip3D.random_flip_data_3d, GlobalRotScaleTrans,
                                                   PointsRangeFilter, ObjectRangeFilter

in the order of the DetMatch pipelines (configs/detmatch/001/detmatch/split_0.py:575-650; shared flip,
then student GlobalRotScaleTrans + range filters and teacher range filter on copies, as
teacher_student_ssl_dataset.py:26-33).  mmcv / mmdet are absent: the registries and helper imports
transforms_3d.py does not use on this path are EMPTY placeholders; mmdet's RandomFlip base class (the 2D
image flip decision) is replaced by an attribute holder and not exercised (do_2d=False, the flip flag
is preset).  PointShuffle is left out of the fixture (torch.randperm order; the kept SET is what counts).
The fixture holds inputs + reference outputs + the recorded img_metas only.

    python tests/golden/gen_pipeline3d_golden.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ssl_geometry_golden import _load, _stub  # noqa: E402


class _Registry(object):
    def register_module(self, *a, **k):
        return lambda cls: cls


class _RandomFlipBase(object):
    def __init__(self, flip_ratio=None, direction='horizontal'):
        self.flip_ratio, self.direction = flip_ratio, direction


def load_reference():
    for n in ('mmdet3d', 'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.core.bbox.structures', 'mmdet3d.ops',
              'mmdet3d.datasets', 'mmdet3d.datasets.pipelines', 'mmcv', 'mmdet', 'mmdet.datasets'):
        _stub(n)
    sys.modules['mmcv'].is_tuple_of = None
    _stub('mmcv.utils', build_from_cfg=None)
    _stub('mmdet3d.ops.roiaware_pool3d', points_in_boxes_gpu=None)
    _stub('mmdet3d.ops.iou3d', iou3d_cuda=None)
    sys.modules['mmdet3d.ops'].roiaware_pool3d = sys.modules['mmdet3d.ops.roiaware_pool3d']
    _stub('mmdet3d.core.points')
    bp = _load('mmdet3d.core.points.base_points', 'mmdet3d/core/points/base_points.py')
    lp = _load('mmdet3d.core.points.lidar_points', 'mmdet3d/core/points/lidar_points.py')
    pts_pkg = sys.modules['mmdet3d.core.points']
    pts_pkg.BasePoints, pts_pkg.LiDARPoints = bp.BasePoints, lp.LiDARPoints
    _load('mmdet3d.core.bbox.structures.utils', 'mmdet3d/core/bbox/structures/utils.py')
    _load('mmdet3d.core.bbox.structures.base_box3d', 'mmdet3d/core/bbox/structures/base_box3d.py')
    lb = _load('mmdet3d.core.bbox.structures.lidar_box3d', 'mmdet3d/core/bbox/structures/lidar_box3d.py')
    sys.modules['mmdet3d.core'].VoxelGenerator = None
    sys.modules['mmdet3d.core.bbox'].box_np_ops = None
    _stub('mmdet.datasets.builder', PIPELINES=_Registry())
    _stub('mmdet.datasets.pipelines', RandomFlip=_RandomFlipBase)
    _stub('mmdet3d.datasets.builder', OBJECTSAMPLERS=_Registry())
    _stub('mmdet3d.datasets.pipelines.data_augment_utils', noise_per_object_v3_=None)
    t3 = _load('mmdet3d.datasets.pipelines.transforms_3d', 'mmdet3d/datasets/pipelines/transforms_3d.py')
    return lp.LiDARPoints, lb.LiDARInstance3DBoxes, t3


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from detmatch_amd import synth
    Points, Boxes, t3 = load_reference()
    pcr = list(synth.KITTI_RANGE)
    out = {'range': np.array(pcr, np.float32)}
    cases = [dict(seed=0, labeled=True, flip=True, std=[0, 0, 0]),
             dict(seed=1, labeled=False, flip=False, std=[0, 0, 0]),
             dict(seed=2, labeled=True, flip=False, std=[0.2, 0.2, 0.1]),
             dict(seed=3, labeled=False, flip=True, std=[0.3, 0.1, 0.05])]
    for ci, c in enumerate(cases):
        frame = synth.lidar_frame(c['seed'])
        pts = frame['points'][::5].copy()                       # ~4000 points, every fifth
        pts[:7, :3] = [[0.0, 0.0, 0.0], [70.4, 0.0, 0.0], [70.39999, 39.99999, 0.99999], [5.0, -40.0, 0.0],
                       [5.0, 5.0, 1.0], [5.0, 5.0, -3.0], [1e-3, -39.999, -2.999]]     # range-boundary rows
        boxes_np, labels_np = synth.frame_to_mm3d_gt(frame)
        d = dict(points=Points(torch.from_numpy(pts.copy()), points_dim=4), bbox3d_fields=[],
                 box_type_3d=Boxes, pcd_horizontal_flip=c['flip'], pcd_vertical_flip=False)
        if c['labeled']:
            d['gt_bboxes_3d'] = Boxes(torch.from_numpy(boxes_np.copy()))
            d['gt_labels_3d'] = labels_np.copy()
            d['bbox3d_fields'].append('gt_bboxes_3d')
        # ---- shared pipeline
        t3.RandomFlip3D(sync_2d=False, do_2d=False, flip_ratio_bev_horizontal=0.5)(d)
        stu, tea = copy.deepcopy(d), d
        # ---- student pipeline
        np.random.seed(100 + ci)
        t3.GlobalRotScaleTrans(rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05],
                               translation_std=c['std'])(stu)
        t3.PointsRangeFilter(point_cloud_range=pcr)(stu)
        if c['labeled']:
            t3.ObjectRangeFilter(point_cloud_range=pcr)(stu)
        # ---- teacher pipeline
        t3.PointsRangeFilter(point_cloud_range=pcr)(tea)
        k = 'c%d_' % ci
        out[k + 'points'] = pts
        out[k + 'flip'] = np.array(c['flip'])
        out[k + 'np_seed'] = np.array(100 + ci)
        out[k + 'std'] = np.array(c['std'], np.float32)
        out[k + 'stu_points'] = stu['points'].tensor.numpy()
        out[k + 'tea_points'] = tea['points'].tensor.numpy()
        out[k + 'stu_rotation'] = np.asarray(stu['pcd_rotation'], dtype=np.float32)
        out[k + 'stu_scale'] = np.array(stu['pcd_scale_factor'], np.float64)
        out[k + 'stu_trans'] = np.asarray(stu['pcd_trans'], np.float64)
        out[k + 'stu_flow'] = np.array(stu['transformation_3d_flow'])
        out[k + 'tea_flow'] = np.array(tea['transformation_3d_flow'] or [''])
        if c['labeled']:
            out[k + 'gt_boxes'], out[k + 'gt_labels'] = boxes_np, labels_np
            out[k + 'stu_boxes'] = stu['gt_bboxes_3d'].tensor.numpy()
            out[k + 'stu_labels'] = np.asarray(stu['gt_labels_3d'])
        print('case %d: %d raw, %d student, %d teacher points, flow %s' % (
            ci, len(pts), len(out[k + 'stu_points']), len(out[k + 'tea_points']), list(out[k + 'stu_flow'])))
    np.savez_compressed(os.path.join(HERE, 'pipeline3d.npz'), **out)
    print('wrote pipeline3d.npz (%d arrays)' % len(out))


if __name__ == '__main__':
    main()
