"""Generates tests/golden/voxelize_*.npz from the REFERENCE hard_voxelize_cpu
(mmdet3d/ops/voxel/src/voxelization_cpu.cpp:105, compiled into oracle/_ref by
oracle/build_ref.py).  Run in the build container only (needs /root/reference):

    python tests/golden/gen_voxelize_golden.py

Fixtures hold inputs + the reference's outputs, nothing else:
  voxelize_kitti000000.npz  the reference test suite's own frame
        tests/data/kitti/training/velodyne_reduced/000000.bin (800 x 4 f32), with the
        parameters of tests/test_models/test_voxel_encoder/test_voxelize.py:15-30
        (voxel 0.5, range [0,-40,-3,70.4,40,1], max_points 1000, max_voxels 20000) and
        with the PV-RCNN config (voxel [.05,.05,.1], max_points 5, max_voxels 16000).
  voxelize_synth.npz        a 3000-point crop of the synthetic KITTI-shaped frame (seed 0)
        with max_voxels = 1500 so that the max_voxels truncation path is exercised.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from detmatch_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
RANGE = [0, -40, -3, 70.4, 40, 1]


def main():
    build_ref.build()
    pts = np.fromfile('/root/reference/tests/data/kitti/training/velodyne_reduced/000000.bin',
                      dtype=np.float32).reshape(-1, 4)
    out = dict(points=pts)
    for tag, vs, mp, mv in (('a', [0.5, 0.5, 0.5], 1000, 20000),
                            ('b', [0.05, 0.05, 0.1], 5, 16000),
                            ('c', [0.5, 0.5, 0.5], 3, 40)):
        v, c, n = build_ref.ref_hard_voxelize(pts, vs, RANGE, mp, mv)
        out['%s_params' % tag] = np.array(vs + RANGE + [mp, mv], np.float64)
        # voxels are stored compacted (only the filled slots) to keep the fixture small
        out['%s_coors' % tag] = c
        out['%s_num' % tag] = n
        out['%s_voxel_sum' % tag] = v.sum(axis=1)
        out['%s_first_pts' % tag] = v[:, 0, :]
        out['%s_last_pts' % tag] = v[np.arange(len(n)), np.maximum(n - 1, 0), :]
    np.savez_compressed(os.path.join(HERE, 'voxelize_kitti000000.npz'), **out)

    p = synth.lidar_frame(0)['points'][:3000]
    v, c, n = build_ref.ref_hard_voxelize(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 1500)
    np.savez_compressed(os.path.join(HERE, 'voxelize_synth.npz'), points=p, voxels=v, coors=c,
                        num=n, params=np.array(list(synth.KITTI_VOXEL) + list(synth.KITTI_RANGE)
                                               + [5, 1500], np.float64))
    print('wrote fixtures; kitti000000 voxels:', {t: len(out['%s_num' % t]) for t in 'abc'},
          'synth voxels:', len(n))


if __name__ == '__main__':
    main()
