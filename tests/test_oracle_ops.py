"""Oracle pinning for rotated IoU / NMS, stacked PointNet++ ops, points-in-boxes.

KATs: SURVEY K4 (compiled reference boxes_iou_bev_cpu), the reference's own
tests/test_utils/test_box3d.py:939-970 (3D IoU), and
tests/test_models/test_common_modules/test_pointnet_ops.py:9-72 (FPS, ball query of the
batch-layout siblings, which share the per-point semantics)."""
import numpy as np
import pytest


def test_k4_bev_iou_kat(orc):
    A = np.array([[0, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, np.pi / 4],
                  [10, 10, 0, 4, 2, 1.5, 0.3]], np.float32)
    B = np.array([[1, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, 0], [10.5, 10.2, 0, 4, 2, 1.5, -0.2],
                  [5, 5, 0, 1, 1, 1, 0]], np.float32)
    got = orc.boxes_iou_bev(A, B)
    want = np.array([[0.3333333, 1.0, 0, 0], [0.2962660, 0.7071069, 0, 0], [0, 0, 0.5505211, 0]],
                    np.float32)
    np.testing.assert_allclose(got, want, rtol=0, atol=6e-8)


def test_reference_boxes3d_overlaps_kat(orc):
    # tests/test_utils/test_box3d.py:948-970; mm3d LiDAR box (x,y,z_bottom,w,l,h,yaw) ->
    # [x, y, z + h/2, w, l, h, -yaw]: the mmdet3d sibling kernel rotates corners clockwise
    # (mmdet3d/ops/iou3d/src/iou3d_kernel.cu:112-118), the pcdet kernel counter-clockwise
    # (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:95-99); everything else is identical
    b1 = np.array([[1.8, -2.5, -1.8, 1.75, 3.39, 1.65, 1.6615927],
                   [8.9, -2.5, -1.6, 1.54, 4.01, 1.57, 1.5215927],
                   [28.3, 0.5, -1.3, 1.47, 2.23, 1.48, 4.7115927],
                   [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35]], np.float32)
    b2 = np.array([[1.2, -3.0, -1.9, 1.8, 3.4, 1.7, 1.9], [8.1, -2.9, -1.8, 1.5, 4.1, 1.6, 1.8],
                   [31.3, -8.2, -1.6, 1.74, 3.77, 1.48, 0.35],
                   [20.1, -28.5, -1.9, 1.6, 3.5, 1.4, 5.1]], np.float32)
    for b in (b1, b2):
        b[:, 2] += b[:, 5] / 2
        b[:, 6] = -b[:, 6]
    want = np.array([[0.3710, 0, 0, 0], [0, 0.3322, 0, 0], [0, 0, 0, 0], [0, 0, 1.0, 0]],
                    np.float32)
    np.testing.assert_allclose(orc.boxes_iou3d(b1, b2), want, rtol=1e-4, atol=1e-7)   # the reference's own tolerance (:968-969)


def test_nms_semantics(orc):
    # three identical boxes + one far away; strict '>' threshold
    b = np.array([[0, 0, 0, 2, 2, 1, 0]] * 3 + [[10, 0, 0, 2, 2, 1, 0]], np.float32)
    assert orc.nms(b, 0.5).tolist() == [0, 3]
    assert orc.nms(b, 1.0).tolist() == [0, 1, 2, 3]      # IoU == 1.0 is not > 1.0
    assert orc.nms(np.zeros((0, 7), np.float32), 0.5).tolist() == []
    rng = np.random.default_rng(0)
    bb = np.concatenate([rng.uniform(0, 20, (200, 2)), np.zeros((200, 1)),
                         rng.uniform(1, 4, (200, 3)), rng.uniform(-3, 3, (200, 1))], 1)
    bb = bb.astype(np.float32)
    keep = orc.nms(bb, 0.1)
    iou = orc.boxes_iou_bev(bb, bb)
    kept = set(keep.tolist())
    for i in range(200):  # greedy definition
        sup = any(iou[j, i] > 0.1 for j in kept if j < i)
        assert (i in kept) == (not sup)


def test_reference_fps_kat(orc):
    xyz = np.array([[[-0.2748, 1.0020, -1.1674], [0.1015, 1.3952, -1.2681],
                     [-0.8070, 2.4137, -0.5845], [-1.0001, 2.1982, -0.5859],
                     [0.3841, 1.8983, -0.7431]],
                    [[-1.0696, 3.0758, -0.1899], [-0.2559, 3.5521, -0.1402],
                     [0.8164, 4.0081, -0.1839], [-1.1000, 3.0213, -0.8205],
                     [-0.0518, 3.7251, -0.3950]]], np.float32)
    assert orc.furthest_point_sample(xyz, 3).tolist() == [[0, 2, 4], [0, 2, 1]]


NEW_XYZ = np.array([[[-0.0740, 1.3147, -1.3625], [-2.2769, 2.7817, -0.2334],
                     [-0.4003, 2.4666, -0.5116], [-0.0740, 1.3147, -1.3625],
                     [-0.0740, 1.3147, -1.3625]],
                    [[-2.0289, 2.4952, -0.1708], [-2.0668, 6.0278, -0.4875],
                     [0.4066, 1.4211, -0.2947], [-2.0289, 2.4952, -0.1708],
                     [-2.0289, 2.4952, -0.1708]]], np.float32)
XYZ = np.array([[[-0.0740, 1.3147, -1.3625], [0.5555, 1.0399, -1.3634],
                 [-0.4003, 2.4666, -0.5116], [-0.5251, 2.4379, -0.8466],
                 [-0.9691, 1.1418, -1.3733], [-0.2232, 0.9561, -1.3626],
                 [-2.2769, 2.7817, -0.2334], [-0.2822, 1.3192, -1.3645],
                 [0.1533, 1.5024, -1.0432], [0.4917, 1.1529, -1.3496]],
                [[-2.0289, 2.4952, -0.1708], [-0.7188, 0.9956, -0.5096],
                 [-2.0668, 6.0278, -0.4875], [-1.9304, 3.3092, 0.6610],
                 [0.0949, 1.4332, 0.3140], [-1.2879, 2.0008, -0.7791],
                 [-0.7252, 0.9611, -0.6371], [0.4066, 1.4211, -0.2947],
                 [0.3220, 1.4447, 0.3548], [-0.9744, 2.3856, -1.2000]]], np.float32)


def test_reference_ball_query_kat(orc):
    idx, empty = orc.ball_query(0.2, 5, XYZ.reshape(-1, 3), [10, 10], NEW_XYZ.reshape(-1, 3),
                                [5, 5])
    want = [[0] * 5, [6] * 5, [2] * 5, [0] * 5, [0] * 5, [0] * 5, [2] * 5, [7] * 5, [0] * 5,
            [0] * 5]
    assert idx.tolist() == want and not empty.any()
    # a wider ball: first-nsample-in-storage-order + padding with the first hit
    idx, _ = orc.ball_query(0.4, 5, XYZ.reshape(-1, 3), [10, 10], NEW_XYZ.reshape(-1, 3), [5, 5])
    assert idx[0].tolist() == [0, 5, 7, 0, 0]
    idx, empty = orc.ball_query(0.05, 4, XYZ.reshape(-1, 3), [10, 10],
                                np.array([[9, 9, 9]], np.float32), [1, 0])
    assert empty.tolist() == [True] and idx.tolist() == [[0, 0, 0, 0]]


GROUP_IDX = np.array([[[0, 0, 0], [3, 3, 3], [8, 8, 8], [0, 0, 0], [0, 0, 0], [0, 0, 0]],
                      [[0, 0, 0], [6, 6, 6], [9, 9, 9], [0, 0, 0], [0, 0, 0], [0, 0, 0]]], np.int32)
GROUP_FEATS = np.array(
    [[[0.5798, -0.7981, -0.9280, -1.3311, 1.3687, 0.9277, -0.4164, -1.8274, 0.9268, 0.8414],
      [5.4247, 1.5113, 2.3944, 1.4740, 5.0300, 5.1030, 1.9360, 2.1939, 2.1581, 3.4666],
      [-1.6266, -1.0281, -1.0393, -1.6931, -1.3982, -0.5732, -1.0830, -1.7561, -1.6786, -1.6967]],
     [[-0.0380, -0.1880, -1.5724, 0.6905, -0.3190, 0.7798, -0.3693, -0.9457, -0.2942, -1.8527],
      [1.1773, 1.5009, 2.6399, 5.9242, 1.0962, 2.7346, 6.0865, 1.5555, 4.3303, 2.8229],
      [-0.6646, -0.6870, -0.1125, -0.2224, -0.3445, -1.4049, 0.4990, -0.7037, -0.9924, 0.0386]]],
    np.float32)
GROUP_PICK = np.array([[[0.5798, -1.3311, 0.9268, 0.5798, 0.5798, 0.5798],
                        [5.4247, 1.4740, 2.1581, 5.4247, 5.4247, 5.4247],
                        [-1.6266, -1.6931, -1.6786, -1.6266, -1.6266, -1.6266]],
                       [[-0.0380, -0.3693, -1.8527, -0.0380, -0.0380, -0.0380],
                        [1.1773, 6.0865, 2.8229, 1.1773, 1.1773, 1.1773],
                        [-0.6646, 0.4990, 0.0386, -0.6646, -0.6646, -0.6646]]], np.float32)


def reference_grouping_kat():
    """tests/test_models/test_common_modules/test_pointnet_ops.py:126-195 (batch layout
    features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample); every expected row repeats
    one value nsample times) re-laid-out for the stacked op: features (B*N, C), idx (B*npoint,
    nsample) local to the sample, out (B*npoint, C, nsample)."""
    feats = np.concatenate([GROUP_FEATS[b].T for b in range(2)])
    idx = GROUP_IDX.reshape(-1, 3)
    want = np.repeat(GROUP_PICK.transpose(0, 2, 1).reshape(12, 3)[:, :, None], 3, axis=2)
    return feats, [10, 10], idx, [6, 6], want


def test_reference_grouping_kat(orc):
    feats, fc, idx, ic, want = reference_grouping_kat()
    assert np.array_equal(orc.group_points(feats, fc, idx, ic), want)


def test_group_points_roundtrip(orc):
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((30, 7)).astype(np.float32)
    idx = rng.integers(0, 10, (12, 4)).astype(np.int32)
    out = orc.group_points(feats, [10, 20], idx, [5, 7])
    for m in range(12):
        start = 0 if m < 5 else 10
        for s in range(4):
            assert np.array_equal(out[m, :, s], feats[start + idx[m, s]])
    g = rng.standard_normal(out.shape).astype(np.float32)
    gf = orc.group_points_grad(g, idx, [5, 7], [10, 20], 30)
    # adjoint: <g, group(f)> == <grad, f>
    np.testing.assert_allclose((g * out).sum(), (gf * feats).sum(), rtol=1e-4)


def test_points_in_boxes_semantics(orc):
    boxes = np.array([[[0, 0, 0, 4, 2, 2, 0.0], [0, 0, 0, 8, 8, 8, 0.0],
                       [10, 0, 0, 4, 2, 2, np.pi / 2]]], np.float32)
    pts = np.array([[[0, 0, 0], [1.9, 0.9, 0.9], [2.5, 0, 0], [0, 0, 1.0], [0, 0, 1.01],
                     [10, 1.9, 0], [10, 2.1, 0], [11.5, 0, 0], [50, 0, 0]]], np.float32)
    got = orc.points_in_boxes(pts, boxes)[0].tolist()
    # first containing box wins; |z - cz| <= dz/2 inclusive; x/y strict with 1e-5 margin;
    # box 2 is rotated by 90 deg so its long side lies along y
    assert got == [0, 0, 1, 0, 1, 2, -1, -1, -1]


# ---------------------------------------------------------------------------------------------
# Reference-compiled / reference-held vectors (round 2)
def test_bev_iou_equals_compiled_reference(orc):
    """tests/golden/iou3d_ref.npz: the reference's own iou3d_cpu.cpp compiled here
    (gen_iou3d_golden.py).  Bit-exact with the FLOAT libm trig calls g++ resolves the reference's
    cos(float) / sin(float) / atan2 to; with the correctly rounded values (default mode, what the
    HIP pre-pass computes; CUDA's device cosf is a third variant) 1 box in ~500 gets a 1-ulp
    different cosine and its IoUs move by <= 2e-6."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'iou3d_ref.npz'))
    assert (g['iou'] > 0).sum() > 300
    with orc.host_libm_trig():
        np.testing.assert_array_equal(orc.boxes_iou_bev(g['a'], g['b']), g['iou'])
    np.testing.assert_allclose(orc.boxes_iou_bev(g['a'], g['b']), g['iou'], rtol=0, atol=3e-6)


def test_nms_keep_equals_reference_greedy(orc):
    """Keep lists = the reference's host greedy pass (iou3d_nms.cpp:117-133) over the compiled
    reference's IoU matrix (the device iou_bev is its line-for-line twin)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'iou3d_ref.npz'))
    boxes = g['nms_boxes']
    with orc.host_libm_trig():
        np.testing.assert_array_equal(orc.boxes_iou_bev(boxes, boxes), g['nms_iou'])
        for thr in (0.01, 0.1, 0.7, 0.8):
            assert np.array_equal(orc.nms(boxes, thr), g['keep_%g' % thr]), thr
    np.testing.assert_allclose(orc.boxes_iou_bev(boxes, boxes), g['nms_iou'], rtol=0, atol=3e-6)
    for thr in (0.01, 0.1, 0.7, 0.8):
        assert np.array_equal(orc.nms(boxes, thr), g['keep_%g' % thr]), thr


def test_reference_points_in_boxes_kat(orc):
    """tests/test_models/test_common_modules/test_roiaware_pool3d.py:43-71 (mmdet3d sibling op:
    boxes (x,y,z_bottom,w,l,h,rz), rotation by rz+pi/2 — points_in_boxes_cuda.cu:24-48) mapped to
    the pcdet convention exactly as the adapter does (openpcdet.py:104-122): centre z, dx=l, dy=w,
    heading = -(rz + pi/2)."""
    boxes = np.array([[[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 0.3]],
                      [[-10.0, 23.0, 16.0, 10, 20, 20, 0.5]]], np.float32)
    pts = np.array([[[1, 2, 3.3], [1.2, 2.5, 3.0], [0.8, 2.1, 3.5], [1.6, 2.6, 3.6],
                     [0.8, 1.2, 3.9], [-9.2, 21.0, 18.2], [3.8, 7.9, 6.3], [4.7, 3.5, -12.2]],
                    [[3.8, 7.6, -2], [-10.6, -12.9, -20], [-16, -18, 9], [-21.3, -52, -5],
                     [0, 0, 0], [6, 7, 8], [-2, -3, -4], [6, 4, 9]]], np.float32)
    pc = boxes.copy()
    pc[..., 2] = boxes[..., 2] + boxes[..., 5] / 2
    pc[..., 3], pc[..., 4] = boxes[..., 4], boxes[..., 3]
    pc[..., 6] = -(boxes[..., 6] + np.float32(np.pi / 2))
    got = orc.points_in_boxes(pts, pc)
    assert got.tolist() == [[0, 0, 0, 0, 0, -1, -1, -1], [-1] * 8]
