"""HIP hard voxelization vs the oracle (bit-exact), through the C-ABI."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RANGE = [0, -40, -3, 70.4, 40, 1]


def _run(dev, pts_list, vs, rng, mp, mv):
    from detmatch_amd import voxel
    t = [torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in pts_list]
    v, c, n, mean, counts = voxel.voxelize_batch(t, vs, rng, mp, mv)
    return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy(), mean.cpu().numpy(), \
        counts.cpu().numpy()


def _check(orc, dev, pts_list, vs, rng, mp, mv):
    v, c, n, mean, counts = _run(dev, pts_list, vs, rng, mp, mv)
    row = 0
    for b, p in enumerate(pts_list):
        ov, oc, on = orc.hard_voxelize(p, vs, rng, mp, mv)
        k = len(on)
        assert counts[b] == k
        assert np.array_equal(c[row:row + k, 0], np.full(k, b))
        assert np.array_equal(c[row:row + k, 1:], oc)     # voxel indices: bit-exact
        assert np.array_equal(n[row:row + k], on)
        assert np.array_equal(v[row:row + k], ov)         # copied floats: bit-exact
        want = ov.sum(axis=1) / np.maximum(on, 1)[:, None].astype(np.float32)
        np.testing.assert_allclose(mean[row:row + k], want, rtol=1e-6, atol=1e-6)
        row += k
    assert counts[len(pts_list)] == row and len(n) == row


def test_survey_kat(orc, dev):
    pts = np.array([(0.5, 0.5, 0.5, 1), (3.5, 3.5, 1.5, 2), (0.6, 0.4, 0.1, 3), (0.7, 0.7, 0.7, 4),
                    (2.5, 0.5, 0.5, 5), (4.0, 1, 1, 6), (-0.1, 1, 1, 7), (2.5, 0.5, 0.6, 8),
                    (3.9999, 3, 1, 9)], np.float32)
    v, c, n, _, counts = _run(dev, [pts], [1, 1, 1], [0, 0, 0, 4, 4, 2], 2, 2)
    assert c[:, 1:].tolist() == [[0, 0, 0], [1, 3, 3]] and n.tolist() == [2, 2]
    assert v[:, :, 3].tolist() == [[1, 3], [2, 9]]


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_golden_reference_frame(orc, dev, tag):
    g = np.load(os.path.join(GOLDEN, 'voxelize_kitti000000.npz'))
    prm = g['%s_params' % tag]
    v, c, n, _, _ = _run(dev, [g['points']], prm[0:3], prm[3:9], int(prm[9]), int(prm[10]))
    assert np.array_equal(c[:, 1:], g['%s_coors' % tag])
    assert np.array_equal(n, g['%s_num' % tag])
    assert np.array_equal(v.sum(axis=1), g['%s_voxel_sum' % tag])
    assert np.array_equal(v[:, 0, :], g['%s_first_pts' % tag])


def test_golden_synth_max_voxels(dev):
    g = np.load(os.path.join(GOLDEN, 'voxelize_synth.npz'))
    prm = g['params']
    v, c, n, _, _ = _run(dev, [g['points']], prm[0:3], prm[3:9], int(prm[9]), int(prm[10]))
    assert np.array_equal(v, g['voxels']) and np.array_equal(c[:, 1:], g['coors'])
    assert np.array_equal(n, g['num'])


@pytest.mark.parametrize('mv', [16000, 40000, 3000])
def test_kitti_shaped_batch(orc, dev, mv):
    from detmatch_amd import synth
    frames = [synth.lidar_frame(s)['points'] for s in (0, 1, 2)]
    _check(orc, dev, frames, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, mv)


def test_ragged_and_empty_samples(orc, dev):
    from detmatch_amd import synth
    p = synth.lidar_frame(5)['points']
    empty = np.zeros((0, 4), np.float32)
    outside = np.array([[-5, 0, 0, 0], [100, 0, 0, 0]], np.float32)
    _check(orc, dev, [p[:10], empty, p[:2000], outside, p[:1]], synth.KITTI_VOXEL,
           synth.KITTI_RANGE, 5, 100)
    _check(orc, dev, [empty], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 100)


def test_division_edge_values(orc, dev):
    # values where floor(x / 0.05f) != floor(x * 20.0f)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 70.4, size=200000).astype(np.float32)
    bad = x[np.floor(x / np.float32(0.05)) != np.floor(x * np.float32(20.0))]
    assert len(bad) > 0
    pts = np.zeros((len(bad) + 2, 4), np.float32)
    pts[:len(bad), 0] = bad
    pts[-2] = [56.699997, 0, 0.8999998, 0]
    pts[-1] = [34.1, 0, 0, 0]
    _check(orc, dev, [pts], [0.05, 0.05, 0.1], RANGE, 5, 20000)


def test_single_sample_module_api(orc, dev):
    from detmatch_amd import synth
    from detmatch_amd.voxel import Voxelization
    m = Voxelization(list(synth.KITTI_VOXEL), list(synth.KITTI_RANGE), 5, (16000, 40000))
    p = synth.lidar_frame(2)['points']
    v, c, n = m(torch.from_numpy(p).to(dev))
    ov, oc, on = orc.hard_voxelize(p, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    assert c.dtype == torch.int32 and c.shape[1] == 3
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(v.cpu().numpy(), ov)
    assert np.array_equal(n.cpu().numpy(), on)


def test_waymo_shaped_full_size(orc, dev):
    """BASELINE config #5 size: ~200k points, grid 1504x1504x40, cap 150000."""
    from detmatch_amd import synth
    p = synth.lidar_frame(0, full360=True)['points']
    assert p.shape[0] > 150000
    _check(orc, dev, [p], synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
