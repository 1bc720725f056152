"""Oracle pinning, hard voxelization: golden vectors from the compiled reference
(tests/golden/gen_voxelize_golden.py), SURVEY K1 KAT, and — when oracle/_ref is
present — the compiled reference itself on fresh inputs."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

RANGE = [0, -40, -3, 70.4, 40, 1]


def test_survey_k1_kat(orc):
    # SURVEY.md §2.3 K1 "Known-answer test produced by the compiled reference hard_voxelize"
    pts = np.array([(0.5, 0.5, 0.5, 1), (3.5, 3.5, 1.5, 2), (0.6, 0.4, 0.1, 3), (0.7, 0.7, 0.7, 4),
                    (2.5, 0.5, 0.5, 5), (4.0, 1, 1, 6), (-0.1, 1, 1, 7), (2.5, 0.5, 0.6, 8),
                    (3.9999, 3, 1, 9)], np.float32)
    v, c, n = orc.hard_voxelize(pts, [1, 1, 1], [0, 0, 0, 4, 4, 2], 2, 2)
    assert c.tolist() == [[0, 0, 0], [1, 3, 3]]
    assert n.tolist() == [2, 2]
    assert v[:, :, 3].tolist() == [[1, 3], [2, 9]]


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_golden_kitti000000(orc, tag):
    g = np.load(os.path.join(GOLDEN, 'voxelize_kitti000000.npz'))
    prm = g['%s_params' % tag]
    v, c, n = orc.hard_voxelize(g['points'], prm[0:3], prm[3:9], int(prm[9]), int(prm[10]))
    assert np.array_equal(c, g['%s_coors' % tag])
    assert np.array_equal(n, g['%s_num' % tag])
    assert np.array_equal(v.sum(axis=1), g['%s_voxel_sum' % tag])
    assert np.array_equal(v[:, 0, :], g['%s_first_pts' % tag])
    assert np.array_equal(v[np.arange(len(n)), np.maximum(n - 1, 0), :], g['%s_last_pts' % tag])


def test_golden_synth_max_voxels(orc):
    g = np.load(os.path.join(GOLDEN, 'voxelize_synth.npz'))
    prm = g['params']
    v, c, n = orc.hard_voxelize(g['points'], prm[0:3], prm[3:9], int(prm[9]), int(prm[10]))
    assert len(n) == 1500  # truncated by max_voxels
    assert np.array_equal(v, g['voxels']) and np.array_equal(c, g['coors'])
    assert np.array_equal(n, g['num'])


def test_division_not_reciprocal(orc):
    # floor(x / 0.05f) != floor(x * 20.0f) for x = 56.699997f -> 1133
    pts = np.array([[56.699997, 0.0, 0.0, 0.0], [34.1, 0.0, 0.8999998, 0.0]], np.float32)
    _, c, _ = orc.hard_voxelize(pts, [0.05, 0.05, 0.1], RANGE, 5, 100)
    assert c[0, 2] == 1133
    assert c[1, 2] == 681 and c[1, 0] == 38


def test_against_compiled_reference(orc):
    from oracle import build_ref
    if build_ref.so_path() is None:
        pytest.skip('oracle/_ref not built (no /root/reference here)')
    from detmatch_amd import synth
    pts = synth.lidar_frame(3)['points']
    for mp, mv in ((5, 16000), (5, 2000), (1, 40000)):
        r = build_ref.ref_hard_voxelize(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, mv)
        o = orc.hard_voxelize(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, mp, mv)
        for a, b in zip(r, o):
            assert np.array_equal(a, b)


def test_empty_and_all_out_of_range(orc):
    v, c, n = orc.hard_voxelize(np.zeros((0, 4), np.float32), [0.05, 0.05, 0.1], RANGE, 5, 100)
    assert len(n) == 0
    pts = np.array([[-1, 0, 0, 0], [80, 0, 0, 0], [1, 0, 5, 0]], np.float32)
    v, c, n = orc.hard_voxelize(pts, [0.05, 0.05, 0.1], RANGE, 5, 100)
    assert len(n) == 0
