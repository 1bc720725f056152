"""Host-side SSL geometry vs goldens produced by the reference's own files
(tests/golden/gen_ssl_geometry_golden.py): LiDARInstance3DBoxes, 3D aug replay / reverse,
3D->2D projection + validity mask, 2D scale/flip transform.  fp32, tolerance 1e-5 abs."""
import os

import numpy as np
import torch

from conftest import GOLDEN

G = np.load(os.path.join(GOLDEN, 'ssl_geometry.npz'))
TOL = dict(rtol=1e-5, atol=1e-4)


def test_box_structure():
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    b = LiDARInstance3DBoxes(torch.from_numpy(G['boxes']))
    np.testing.assert_allclose(b.corners.numpy(), G['corners'], **TOL)
    np.testing.assert_allclose(b.gravity_center.numpy(), G['gravity_center'], **TOL)
    g = LiDARInstance3DBoxes(torch.from_numpy(G['boxes']), origin=(0.5, 0.5, 0.5))
    np.testing.assert_allclose(g.tensor.numpy(), G['from_gravity_origin'], **TOL)
    assert len(b[2:5]) == 3 and len(b[3]) == 1 and len(b[torch.tensor([True] * 24)]) == 24


def test_projection_and_validity():
    from detmatch_amd.mm3d.bbox_utils import bbox_3d_to_bbox_2d
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    b = LiDARInstance3DBoxes(torch.from_numpy(G['boxes']))
    xyxy, valid = bbox_3d_to_bbox_2d(b, G['lidar2img'], (375, 1242, 3))
    assert np.array_equal(valid.numpy(), G['proj_valid'])          # mask: exact
    np.testing.assert_allclose(xyxy.numpy(), G['proj_xyxy'], rtol=1e-5, atol=2e-3)
    e, v = bbox_3d_to_bbox_2d(LiDARInstance3DBoxes(torch.zeros((0, 7))), G['lidar2img'], (375, 1242, 3))
    assert e.shape == (0, 4) and v.shape == (0,)
    # autograd flows through the projection (Bboxes3DTo2D is differentiable)
    t = torch.from_numpy(G['boxes']).clone().requires_grad_(True)
    xy, _ = bbox_3d_to_bbox_2d(LiDARInstance3DBoxes(t) if False else _wrap(t), G['lidar2img'],
                               (375, 1242, 3))
    xy.sum().backward()
    assert t.grad is not None and torch.isfinite(t.grad).all()


def _wrap(t):
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    b = LiDARInstance3DBoxes(t.detach())
    b.tensor = t
    return b


def test_3d_augmentation_replay():
    from detmatch_amd.mm3d.bbox_utils import apply_3d_transformation_bboxes
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    meta = dict(pcd_rotation=torch.from_numpy(G['meta_rotation']),
                pcd_scale_factor=float(G['meta_scale']), pcd_trans=G['meta_trans'],
                pcd_horizontal_flip=True, pcd_vertical_flip=False,
                transformation_3d_flow=['HF', 'R', 'S', 'T'])
    b = LiDARInstance3DBoxes(torch.from_numpy(G['boxes']))
    fwd = apply_3d_transformation_bboxes(b, meta, reverse=False)
    np.testing.assert_allclose(fwd.tensor.numpy(), G['aug_forward'], **TOL)
    back = apply_3d_transformation_bboxes(fwd, meta, reverse=True)
    np.testing.assert_allclose(back.tensor.numpy(), G['aug_roundtrip'], **TOL)
    np.testing.assert_allclose(b.tensor.numpy(), G['boxes'])       # input untouched


def test_2d_transform():
    from detmatch_amd.mm3d.bbox_utils import bbox_2d_transform
    meta = dict(img_shape=(384, 1272, 3), ori_shape=(375, 1242, 3), scale_factor=G['meta2_scale'],
                flip=True)
    o2n = bbox_2d_transform(meta, torch.from_numpy(G['boxes2d']), True)
    np.testing.assert_allclose(o2n.numpy(), G['boxes2d_ori2new'], **TOL)
    back = bbox_2d_transform(meta, o2n, False)
    np.testing.assert_allclose(back.numpy(), G['boxes2d_back'], **TOL)
    np.testing.assert_allclose(back.numpy(), G['boxes2d'], rtol=1e-4, atol=1e-2)


def test_mlvl_helpers():
    from detmatch_amd.mm3d.bbox_utils import mlvl_get, mlvl_set
    import pytest
    d = {}
    mlvl_set(d, 'a.b.c', 1)
    assert mlvl_get(d, 'a.b.c') == 1 and mlvl_get(d, 'a.x.c', 5) == 5
    with pytest.raises(Exception):
        mlvl_set(d, 'a.b.c', 2)


def test_mm3d_pcdet_conversion_keeps_the_box_geometry():
    """openpcdet.py:100-122 / :211-224: the converted box covers the same eight corners.  Both corner
    functions are pinned to the reference's outputs (LiDARInstance3DBoxes.corners in ssl_geometry.npz,
    boxes_to_corners_3d in pcdet_torch.npz), so this pins the conversion formulas geometrically."""
    import numpy as np
    import torch
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    from detmatch_amd.mm3d.openpcdet import mm3d_to_pcdet_boxes, pcdet_to_mm3d_boxes
    from detmatch_amd.pcdet.utils import boxes_to_corners_3d
    rng = np.random.default_rng(0)
    b = np.stack([rng.uniform(0, 70, 30), rng.uniform(-40, 40, 30), rng.uniform(-2, 0, 30), rng.uniform(0.5, 2, 30),
                  rng.uniform(0.6, 5, 30), rng.uniform(1, 2, 30), rng.uniform(-6, 6, 30)], 1).astype(np.float32)
    mm = LiDARInstance3DBoxes(torch.from_numpy(b))
    pc = mm3d_to_pcdet_boxes(mm)
    srt = lambda c: np.sort(np.round(c.reshape(30, 8, 3).numpy().astype(np.float64), 3).view([('', np.float64)] * 3),
                            axis=1).view(np.float64).reshape(30, 8, 3)
    np.testing.assert_allclose(srt(boxes_to_corners_3d(pc)), srt(mm.corners), atol=2e-3)
    back = pcdet_to_mm3d_boxes(pc)
    np.testing.assert_allclose(srt(back.corners), srt(mm.corners), atol=2e-3)
    assert float(pc[:, 6].min()) >= -np.pi - 1e-5 and float(pc[:, 6].max()) < np.pi + 1e-5   # limit_period(., .5, 2 pi)


def test_composed_3d_transformation_is_the_replayed_ops():
    """compose_3d_transformation (the affine map the device projection kernel takes) applied with numpy
    == the REFERENCE's op-by-op replay (goldens of apply_3d_transformation_bboxes), both directions,
    plus every flow order / flip combination against the replay itself."""
    from detmatch_amd.mm3d.bbox_utils import apply_3d_transformation_bboxes, compose_3d_transformation
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes

    def apply(boxes, meta, reverse):
        A, t, s, sigma, off = compose_3d_transformation(meta, reverse=reverse)
        out = boxes.astype(np.float64).copy()
        out[:, :3] = out[:, :3] @ A + t
        out[:, 3:6] *= s
        out[:, 6] = sigma * out[:, 6] + off
        return out

    meta = dict(pcd_rotation=torch.from_numpy(G['meta_rotation']),
                pcd_scale_factor=float(G['meta_scale']), pcd_trans=G['meta_trans'],
                pcd_horizontal_flip=True, pcd_vertical_flip=False,
                transformation_3d_flow=['HF', 'R', 'S', 'T'])
    np.testing.assert_allclose(apply(G['boxes'], meta, False), G['aug_forward'], **TOL)
    np.testing.assert_allclose(apply(G['aug_forward'], meta, True), G['aug_roundtrip'], **TOL)
    rng = np.random.default_rng(0)
    for flow, hf, vf in ((['T', 'S', 'R', 'VF', 'HF'], True, True), (['R', 'VF'], False, True),
                         (['HF', 'VF', 'R', 'S', 'T'], True, False), ([], False, False), (['S'], False, False)):
        ang = rng.uniform(-0.7, 0.7)
        m = dict(pcd_rotation=torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0],
                                            [0, 0, 1]], dtype=torch.float32),
                 pcd_scale_factor=float(rng.uniform(0.9, 1.1)), pcd_trans=rng.normal(0, 0.3, 3).astype(np.float32),
                 pcd_horizontal_flip=hf, pcd_vertical_flip=vf, transformation_3d_flow=flow)
        for reverse in (False, True):
            want = apply_3d_transformation_bboxes(LiDARInstance3DBoxes(torch.from_numpy(G['boxes'])), m,
                                                  reverse=reverse).tensor.numpy()
            np.testing.assert_allclose(apply(G['boxes'], m, reverse), want, **TOL)


def test_step_generators_are_driven_in_lockstep():
    """spconv.ops.drive_steps_together: every round's device scalars come back in one copy, each generator
    is sent its own value and their return values keep the job order; drive_steps runs one alone."""
    from detmatch_amd.spconv.ops import drive_steps, drive_steps_together
    log = []

    def job(name, asks):
        got = []
        for a in asks:
            log.append((name, a))
            got.append((yield torch.tensor([a], dtype=torch.int32)))
        return name, got

    res = drive_steps_together([job('a', [3, 5, 7]), job('b', [11]), job('c', []), job('d', [2, 4])])
    assert res == [('a', [3, 5, 7]), ('b', [11]), ('c', []), ('d', [2, 4])]
    assert log == [('a', 3), ('b', 11), ('d', 2), ('a', 5), ('d', 4), ('a', 7)]      # round by round
    assert drive_steps(job('e', [9, 8])) == ('e', [9, 8])
