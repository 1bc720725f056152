"""Generates tests/golden/dbsampler/ (a small synthetic GT database in the KITTI db-info format) and
tests/golden/dbsampler.npz (outputs of the REFERENCE's own DataBaseSampler on it; build container only):

    mmdet3d/datasets/pipelines/dbsampler.py            BatchSampler, DataBaseSampler.sample_all /
                                                       sample_class_v2 / filter_by_*
    mmdet3d/datasets/pipelines/data_augment_utils.py   box_collision_test
    mmdet3d/core/bbox/box_np_ops.py                    center_to_corner_box2d, ...
    mmdet3d/core/points/{base_points,lidar_points}.py  LiDARPoints (the loaded object points)

loaded by file path.  numba is absent: its decorators are the identity here (the decorated functions are
plain numpy loops).  mmcv / mmdet / cv2 are absent: the registries are EMPTY placeholders, `mmcv.load`
is pickle, and the points loader the sampler builds from its config is a 4-line reader of the `.bin`
object files returning the reference's LiDARPoints.  The database itself is synthetic (objects cut out
of detmatch_amd.synth frames).  The fixture holds inputs + reference outputs only.

    python tests/golden/gen_dbsampler_golden.py
"""
import logging
import os
import pickle
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gen_ssl_geometry_golden import _load, _stub  # noqa: E402

CLASSES = ['Pedestrian', 'Cyclist', 'Car']
DB = os.path.join(HERE, 'dbsampler')


def build_database():
    """Objects of four synthetic frames (points inside each GT box, stored relative to the box)."""
    import oracle
    from detmatch_amd import synth
    os.makedirs(os.path.join(DB, 'gt_database'), exist_ok=True)
    infos = {c: [] for c in CLASSES}
    for seed in (10, 11, 12, 13):
        f = synth.lidar_frame(seed)
        boxes, labels = synth.frame_to_mm3d_gt(f)
        idx = oracle.points_in_boxes(f["points"][None, :, :3], f["gt_boxes"][None].astype(np.float32))[0]
        for g in range(len(boxes)):
            pts = f['points'][idx == g].copy()
            if len(pts) < 5 or len(infos[CLASSES[labels[g]]]) >= 6:
                continue
            pts[:, :3] -= boxes[g, :3]
            name = CLASSES[labels[g]]
            rel = 'gt_database/%d_%s_%d.bin' % (seed, name, g)
            pts[:200].astype(np.float32).tofile(os.path.join(DB, rel))
            infos[name].append(dict(name=name, path=rel, image_idx=seed, gt_idx=g, box3d_lidar=boxes[g].astype(np.float32),
                                    num_points_in_gt=int(min(len(pts), 200)), difficulty=int(g % 3) - (g % 5 == 4),
                                    group_id=len(infos[name])))
    with open(os.path.join(DB, 'dbinfos.pkl'), 'wb') as fh:
        pickle.dump(infos, fh)
    return infos


def load_reference():
    nb = types.ModuleType('numba')

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f
    nb.jit = nb.njit = jit
    nb.prange = range
    sys.modules['numba'] = nb
    for n in ('mmdet3d', 'mmdet3d.core', 'mmdet3d.core.bbox', 'mmdet3d.datasets', 'mmdet3d.datasets.pipelines',
              'mmdet', 'mmdet3d.core.points'):
        _stub(n)
    _stub('cv2')
    bp = _load('mmdet3d.core.points.base_points', 'mmdet3d/core/points/base_points.py')
    lp = _load('mmdet3d.core.points.lidar_points', 'mmdet3d/core/points/lidar_points.py')

    class Reg(object):
        def register_module(self, *a, **k):
            return lambda cls: cls

    def loader_from_cfg(cfg, registry):
        def load(results):
            pts = np.fromfile(results['pts_filename'], dtype=np.float32).reshape(-1, cfg['load_dim'])
            return dict(points=lp.LiDARPoints(torch.from_numpy(pts[:, cfg['use_dim']].copy()), points_dim=4))
        return load
    _stub('mmcv', load=lambda p: pickle.load(open(p, 'rb')), build_from_cfg=loader_from_cfg)
    _stub('mmdet3d.utils', get_root_logger=lambda **k: logging.getLogger('ref'))
    _stub('mmdet.datasets', PIPELINES=Reg())
    _stub('mmdet3d.datasets.builder', OBJECTSAMPLERS=Reg())
    ops = _load('mmdet3d.core.bbox.box_np_ops', 'mmdet3d/core/bbox/box_np_ops.py')
    sys.modules['mmdet3d.core.bbox'].box_np_ops = ops
    dau = _load('mmdet3d.datasets.pipelines.data_augment_utils', 'mmdet3d/datasets/pipelines/data_augment_utils.py')
    sys.modules['mmdet3d.datasets.pipelines'].data_augment_utils = dau
    return _load('mmdet3d.datasets.pipelines.dbsampler', 'mmdet3d/datasets/pipelines/dbsampler.py'), ops


def main():
    from detmatch_amd import synth
    infos = build_database()
    db, ops = load_reference()
    f = synth.lidar_frame(0)
    boxes, labels = synth.frame_to_mm3d_gt(f)
    out = dict(scene_boxes=boxes, scene_labels=labels, scene_points=f['points'])
    np.random.seed(5)
    sampler = db.DataBaseSampler(os.path.join(DB, 'dbinfos.pkl'), DB + '/', rate=1.0,
                                 prepare=dict(filter_by_difficulty=[-1],
                                              filter_by_min_points=dict(Car=8, Pedestrian=8, Cyclist=8)),
                                 sample_groups=dict(Car=12, Pedestrian=6, Cyclist=6), classes=CLASSES)
    out['kept_per_class'] = np.array([len(sampler.db_infos[c]) for c in CLASSES])
    for call in range(3):                     # the sampler keeps its cursor between frames
        ret = sampler.sample_all(boxes.copy(), labels.copy())
        k = 'call%d_' % call
        out[k + 'none'] = np.array(ret is None)
        if ret is not None:
            out[k + 'labels'], out[k + 'boxes'] = ret['gt_labels_3d'], ret['gt_bboxes_3d']
            out[k + 'points'], out[k + 'group_ids'] = ret['points'].tensor.numpy(), ret['group_ids']
            # the scene points the reference's ObjectSample would drop (numba points_in_rbbox)
            inside = ops.points_in_rbbox(f['points'][:, :3].astype(np.float32), ret['gt_bboxes_3d'])
            out[k + 'scene_keep'] = np.logical_not(inside.any(-1))
            print('call %d: pasted %d objects %s, %d points; %d scene points removed' % (
                call, len(ret['gt_labels_3d']), ret['gt_labels_3d'].tolist(), len(out[k + 'points']),
                int((~out[k + 'scene_keep']).sum())))
    # road-plane placement (dbsampler.py:192-243) with the calibration of the reference's KITTI fixture frame
    lp = sys.modules['mmdet3d.core.points.lidar_points']
    calib = pickle.load(open('/root/reference/tests/data/kitti/kitti_infos_train.pkl', 'rb'))[0]['calib']
    rb = out['call0_boxes'].copy()
    pts = [lp.LiDARPoints(torch.from_numpy(out['call0_points'][i * 20:(i + 1) * 20].copy()), points_dim=4)
           for i in range(len(rb))]
    plane = np.array([0.01, -0.999, 0.02, 1.62], dtype=np.float32)
    nb, npts = sampler.put_boxes_on_road_planes(rb, pts, dict(calib=calib, road_plane=plane))
    out.update(rp_plane=plane, rp_boxes=nb, rp_points=np.concatenate([p.tensor.numpy() for p in npts]),
               rp_R0=calib['R0_rect'], rp_Tr=calib['Tr_velo_to_cam'])
    np.savez_compressed(os.path.join(HERE, 'dbsampler.npz'), **out)
    print('database: %s objects; wrote dbsampler.npz' % {c: len(v) for c, v in infos.items()})


if __name__ == '__main__':
    main()
