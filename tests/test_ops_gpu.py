"""HIP rotated IoU / NMS, stacked PointNet++ ops and points-in-boxes vs the oracle,
through the C-ABI.  Index outputs (NMS keep lists, ball-query indices, FPS indices,
box ids) are compared bit for bit; IoU values too (same fp32 op sequence, transcendentals
"double libm rounded to float" on both sides)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(rng, n, spread=30.0):
    return np.concatenate([rng.uniform(0, spread, (n, 2)), rng.uniform(-1, 1, (n, 1)),
                           rng.uniform(0.5, 5, (n, 3)), rng.uniform(-3.2, 3.2, (n, 1))],
                          1).astype(np.float32)


def test_iou_matrices(orc, dev):
    from detmatch_amd import iou3d_nms
    rng = np.random.default_rng(0)
    a, b = _boxes(rng, 150), _boxes(rng, 97)
    a[3] = b[5]                      # an identical pair
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    assert np.array_equal(iou3d_nms.boxes_iou_bev(ta, tb).cpu().numpy(), orc.boxes_iou_bev(a, b))
    assert np.array_equal(iou3d_nms.boxes_overlap_bev(ta, tb).cpu().numpy(),
                          orc.boxes_overlap_bev(a, b))
    np.testing.assert_allclose(iou3d_nms.boxes_iou3d_gpu(ta, tb).cpu().numpy(),
                               orc.boxes_iou3d(a, b), rtol=1e-6, atol=1e-7)
    # SURVEY K4 KAT
    A = torch.tensor([[0, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, np.pi / 4],
                      [10, 10, 0, 4, 2, 1.5, 0.3]], dtype=torch.float32, device=dev)
    B = torch.tensor([[1, 0, 0, 2, 2, 1, 0], [0, 0, 0, 2, 2, 1, 0], [10.5, 10.2, 0, 4, 2, 1.5, -0.2],
                      [5, 5, 0, 1, 1, 1, 0]], dtype=torch.float32, device=dev)
    want = np.array([[0.3333333, 1.0, 0, 0], [0.2962660, 0.7071069, 0, 0], [0, 0, 0.5505211, 0]],
                    np.float32)
    np.testing.assert_allclose(iou3d_nms.boxes_iou_bev(A, B).cpu().numpy(), want, atol=6e-8)
    assert iou3d_nms.boxes_iou_bev(A[:0], B).shape == (0, 4)


@pytest.mark.parametrize('n,thresh,spread', [(1, 0.5, 30), (63, 0.1, 10), (64, 0.7, 10),
                                             (65, 0.01, 10), (1000, 0.7, 40), (4096, 0.8, 60),
                                             (9000, 0.8, 120)])
def test_nms_keep_lists(orc, dev, n, thresh, spread):
    from detmatch_amd import iou3d_nms
    rng = np.random.default_rng(n)
    b = _boxes(rng, n, spread)
    scores = rng.permutation(n).astype(np.float32)    # distinct scores: no tie ambiguity
    order = np.argsort(-scores, kind='stable')
    want = order[orc.nms(b[order], thresh)]
    got, _ = iou3d_nms.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(scores).to(dev),
                               thresh)
    assert np.array_equal(got.cpu().numpy(), want)      # keep mask: bit-exact
    got2, _ = iou3d_nms.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(scores).to(dev),
                                thresh, pre_maxsize=min(n, 512), post_max_size=100)
    w2 = order[:512][orc.nms(b[order[:512]], thresh)][:100]
    assert np.array_equal(got2.cpu().numpy(), w2)
    want_n = order[orc.nms(b[order], thresh, normal=True)]
    got_n, _ = iou3d_nms.nms_normal_gpu(torch.from_numpy(b).to(dev),
                                        torch.from_numpy(scores).to(dev), thresh)
    assert np.array_equal(got_n.cpu().numpy(), want_n)


@pytest.mark.parametrize('n,budget,spread', [(3000, 300, 4.0), (5000, 400, 5.0), (9000, 512, 60.0)])
def test_nms_two_phase_budget(orc, dev, n, budget, spread):
    """With a survivor budget the mask is built for the leading blocks first; crowded scenes
    (small spread: few survivors among the first 1024 boxes) force the second phase to resume."""
    from detmatch_amd import iou3d_nms
    rng = np.random.default_rng(n + budget)
    b = _boxes(rng, n, spread)
    scores = rng.permutation(n).astype(np.float32)
    order = np.argsort(-scores, kind='stable')
    full = order[orc.nms(b[order], 0.3)]
    want = full[:budget]
    got, _ = iou3d_nms.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(scores).to(dev), 0.3,
                               post_max_size=budget)
    assert np.array_equal(got.cpu().numpy(), want)
    lead_hits = int((np.isin(order[:max(4 * budget, 1024)], full)).sum())
    if spread <= 6.0:
        assert lead_hits < budget < n  # the case really exercises the resume path
        assert len(full) > lead_hits   # ... and finds more survivors there


def test_nms_empty(dev):
    from detmatch_amd import iou3d_nms
    k, _ = iou3d_nms.nms_gpu(torch.zeros((0, 7), device=dev), torch.zeros((0,), device=dev), 0.5)
    assert k.shape == (0,)


def _stacked(rng, counts, lo, hi):
    return np.concatenate([rng.uniform(lo, hi, (c, 3)) for c in counts]).astype(np.float32)


@pytest.mark.parametrize('radius,nsample', [(0.4, 16), (0.8, 16), (1.2, 32), (4.0, 16)])
def test_ball_query_and_group(orc, dev, radius, nsample):
    from detmatch_amd import pointnet2_stack as pn
    rng = np.random.default_rng(int(radius * 10))
    xyz_cnt, new_cnt = [3000, 0, 1700], [253, 0, 300]      # sample boundary inside a block of queries
    xyz = _stacked(rng, xyz_cnt, 0, 12)
    new_xyz = _stacked(rng, new_cnt, -1, 13)
    feats = rng.standard_normal((xyz.shape[0], 19)).astype(np.float32)
    t = lambda a, dt=None: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    idx, empty = pn.ball_query(radius, nsample, t(xyz), t(xyz_cnt, np.int32), t(new_xyz),
                               t(new_cnt, np.int32))
    oidx, oempty = orc.ball_query(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(empty.cpu().numpy(), oempty)
    tf = t(feats).requires_grad_(True)
    g = pn.grouping_operation(tf, t(xyz_cnt, np.int32), idx, t(new_cnt, np.int32))
    og = orc.group_points(feats, xyz_cnt, oidx, new_cnt)
    assert np.array_equal(g.detach().cpu().numpy(), og)
    gz = pn.grouping_operation(tf, t(xyz_cnt, np.int32), idx, t(new_cnt, np.int32), empty)
    og[oempty] = 0
    assert np.array_equal(gz.detach().cpu().numpy(), og)
    dy = rng.standard_normal(og.shape).astype(np.float32)
    g.backward(t(dy))
    want = orc.group_points_grad(dy, oidx, new_cnt, xyz_cnt, xyz.shape[0])
    np.testing.assert_allclose(tf.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('counts', [([5000, 2300], [9001, 8999]), ([1, 1024, 1025], [3, 5, 2]),
                                    ([700, 64], [1, 2047])])
def test_ball_query_tile_and_wave_paths(orc, dev, counts):
    """Both launch shapes (4 queries per wave above 16 k queries), tiles that end exactly on / one past
    the 1024-point staging tile, one-point samples."""
    from detmatch_amd import pointnet2_stack as pn
    xyz_cnt, new_cnt = counts
    rng = np.random.default_rng(sum(xyz_cnt))
    xyz = _stacked(rng, xyz_cnt, 0, 6)
    new_xyz = _stacked(rng, new_cnt, -0.5, 6.5)
    t = lambda a, dt=None: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    for radius, nsample in ((0.3, 16), (0.9, 32)):
        idx, empty = pn.ball_query(radius, nsample, t(xyz), t(xyz_cnt, np.int32), t(new_xyz),
                                   t(new_cnt, np.int32))
        oidx, oempty = orc.ball_query(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt)
        assert np.array_equal(idx.cpu().numpy(), oidx)
        assert np.array_equal(empty.cpu().numpy(), oempty)


def test_reference_ball_query_kat(dev):
    from detmatch_amd import pointnet2_stack as pn
    from test_oracle_ops import NEW_XYZ, XYZ
    cnt = torch.tensor([10, 10], dtype=torch.int32, device=dev)
    ncnt = torch.tensor([5, 5], dtype=torch.int32, device=dev)
    idx, empty = pn.ball_query(0.2, 5, torch.from_numpy(XYZ.reshape(-1, 3)).to(dev), cnt,
                               torch.from_numpy(NEW_XYZ.reshape(-1, 3)).to(dev), ncnt)
    want = [[0] * 5, [6] * 5, [2] * 5, [0] * 5, [0] * 5, [0] * 5, [2] * 5, [7] * 5, [0] * 5,
            [0] * 5]
    assert idx.cpu().tolist() == want and not bool(empty.any())


def test_group_large_channels(orc, dev):
    """RoI-grid shape: C = 128, nsample = 16."""
    from detmatch_amd import pointnet2_stack as pn
    rng = np.random.default_rng(5)
    feats = rng.standard_normal((4096, 128)).astype(np.float32)
    idx = rng.integers(0, 2048, (3000, 16)).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    cnt = torch.tensor([2048, 2048], dtype=torch.int32, device=dev)
    icnt = torch.tensor([1000, 2000], dtype=torch.int32, device=dev)
    g = pn.grouping_operation(t(feats), cnt, t(idx), icnt)
    assert np.array_equal(g.cpu().numpy(), orc.group_points(feats, [2048, 2048], idx, [1000, 2000]))


@pytest.mark.parametrize('c,ns,n_src,local', [(128, 16, 4096, True), (128, 16, 4096, False),
                                               (64, 32, 60000, False), (16, 16, 4096, True)])
def test_group_grad_combining_kernel(orc, dev, c, ns, n_src, local):
    """m >= 16384 takes the LDS-combining backward (on-chip sum of everything that shares a
    source row, one flush per distinct row); `local` = RoI-grid-like overlap (few distinct rows
    per chunk), not local = every reference distinct (table overflow path)."""
    from detmatch_amd import pointnet2_stack as pn
    rng = np.random.default_rng(c + ns)
    m, half = 20000, n_src // 2
    if local:
        centre = rng.integers(0, half - 40, (m // 200 + 1,)).repeat(200)[:m]
        idx = (centre[:, None] + rng.integers(0, 40, (m, ns))).astype(np.int32)
    else:
        idx = rng.integers(0, half, (m, ns)).astype(np.int32)
    idx[::7, 3:] = idx[::7, :1]           # ball-query padding: repeats of the first neighbour
    feats = torch.from_numpy(rng.standard_normal((n_src, c)).astype(np.float32)).to(dev).requires_grad_()
    cnt = torch.tensor([half, n_src - half], dtype=torch.int32, device=dev)
    icnt = torch.tensor([12000, 8000], dtype=torch.int32, device=dev)
    out = pn.grouping_operation(feats, cnt, torch.from_numpy(idx).to(dev), icnt)
    g = rng.standard_normal((m, c, ns)).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    want = orc.group_points_grad(g, idx, [12000, 8000], [half, n_src - half], n_src)
    np.testing.assert_allclose(feats.grad.cpu().numpy(), want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('n,m', [(5, 3), (100, 37), (1024, 512), (3000, 2048), (19940, 2048)])
def test_fps(orc, dev, n, m):
    from detmatch_amd import pointnet2_stack as pn
    from detmatch_amd import synth
    if n == 19940:
        fr = [synth.lidar_frame(s)['points'][:, :3] for s in (0, 1)]
        n = min(len(f) for f in fr)
        xyz = np.stack([f[:n] for f in fr])
    else:
        xyz = np.random.default_rng(n).uniform(-20, 20, (3, n, 3)).astype(np.float32)
    got = pn.furthest_point_sample(torch.from_numpy(np.ascontiguousarray(xyz)).to(dev), m)
    assert np.array_equal(got.cpu().numpy(), orc.furthest_point_sample(xyz, m))


def test_fps_reference_kat_and_ties(orc, dev):
    from detmatch_amd import pointnet2_stack as pn
    xyz = np.array([[[-0.2748, 1.0020, -1.1674], [0.1015, 1.3952, -1.2681],
                     [-0.8070, 2.4137, -0.5845], [-1.0001, 2.1982, -0.5859],
                     [0.3841, 1.8983, -0.7431]],
                    [[-1.0696, 3.0758, -0.1899], [-0.2559, 3.5521, -0.1402],
                     [0.8164, 4.0081, -0.1839], [-1.1000, 3.0213, -0.8205],
                     [-0.0518, 3.7251, -0.3950]]], np.float32)
    assert pn.furthest_point_sample(torch.from_numpy(xyz).to(dev), 3).cpu().tolist() == \
        [[0, 2, 4], [0, 2, 1]]
    # duplicated points -> exact distance ties; the block-size dependent tie rule must match
    rng = np.random.default_rng(9)
    base = rng.uniform(-5, 5, (1, 700, 3)).astype(np.float32)
    dup = np.concatenate([base, base, base[:, :300]], 1)
    got = pn.furthest_point_sample(torch.from_numpy(dup).to(dev), 64)
    assert np.array_equal(got.cpu().numpy(), orc.furthest_point_sample(dup, 64))


@pytest.mark.parametrize('n,m', [(700, 64), (5000, 512), (24000, 300)])
def test_fps_stack_ragged_with_duplicates_equals_oracle(orc, dev, n, m):
    """The one-workgroup kernel (packed distance arithmetic, value-only reductions, index recovered by an LDS key
    minimum) on ragged stacked samples with duplicated points (exact ties decided by the reference's stride-class
    rule): the oracle's indices bit for bit, sample by sample."""
    from detmatch_amd import pointnet2_stack as pn2
    g = torch.Generator().manual_seed(n + m)
    sizes = [n, n - 13]
    pts = [torch.rand(s_, 3, generator=g) * torch.tensor([70.0, 80.0, 4.0]) for s_ in sizes]
    pts[0][50:150] = pts[0][300:400]
    pts[1][:40] = pts[1][200:240]
    xyz = torch.cat(pts).to(dev).contiguous()
    got = pn2.furthest_point_sample_stack(xyz, torch.tensor(sizes, dtype=torch.int32), m).cpu().numpy()
    for b, p in enumerate(pts):
        want = orc.furthest_point_sample(p.numpy()[None], m)[0]
        assert np.array_equal(got[b], want), b


def test_points_in_boxes(orc, dev):
    from detmatch_amd import roiaware_pool3d
    rng = np.random.default_rng(2)
    boxes = np.stack([_boxes(rng, 30, 20.0) for _ in range(2)])
    boxes[1, 20:] = 0          # zero-padded GT rows
    pts = rng.uniform(-2, 22, (2, 2048, 3)).astype(np.float32)
    pts[..., 2] = rng.uniform(-2, 2, (2, 2048))
    got = roiaware_pool3d.points_in_boxes_gpu(torch.from_numpy(pts).to(dev),
                                              torch.from_numpy(boxes).to(dev))
    want = orc.points_in_boxes(pts, boxes)
    assert np.array_equal(got.cpu().numpy(), want)
    assert (want >= 0).sum() > 50
    b2 = np.array([[[0, 0, 0, 4, 2, 2, 0.0], [0, 0, 0, 8, 8, 8, 0.0],
                    [10, 0, 0, 4, 2, 2, np.pi / 2]]], np.float32)
    p2 = np.array([[[0, 0, 0], [1.9, 0.9, 0.9], [2.5, 0, 0], [0, 0, 1.0], [0, 0, 1.01],
                    [10, 1.9, 0], [10, 2.1, 0], [11.5, 0, 0], [50, 0, 0]]], np.float32)
    got = roiaware_pool3d.points_in_boxes_gpu(torch.from_numpy(p2).to(dev),
                                              torch.from_numpy(b2).to(dev))
    assert got.cpu().tolist() == [[0, 0, 1, 0, 1, 2, -1, -1, -1]]


@pytest.mark.parametrize('m_per,c,with_feats', [(300, 16, True), (9000, 32, True), (500, 0, False)])
def test_sa_module_row_layout_equals_reference_formulation(dev, m_per, c, with_feats):
    """StackSAModuleMSG in row layout (fused gather, GEMM MLP, column BatchNorm) == the reference's
    (1, C, M, nsample) Conv2d formulation built from the oracle-checked group ops: outputs, input
    gradient, weight gradients and BatchNorm running statistics."""
    import copy
    from detmatch_amd import pointnet2_stack as pn
    rng = np.random.default_rng(m_per + c)
    n = [1500, 1100]
    xyz = torch.from_numpy(rng.uniform(-4, 4, (sum(n), 3)).astype(np.float32)).to(dev)
    new_xyz = torch.from_numpy(rng.uniform(-5, 5, (2 * m_per, 3)).astype(np.float32)).to(dev)
    cnt = torch.tensor(n, dtype=torch.int32, device=dev)
    ncnt = torch.tensor([m_per, m_per], dtype=torch.int32, device=dev)
    torch.manual_seed(0)
    a = pn.StackSAModuleMSG(radii=[0.6, 1.5], nsamples=[8, 16], mlps=[[c, 16, 24], [c, 16, 32]],
                            use_xyz=True).to(dev)
    b = copy.deepcopy(a)
    b.row_layout = False
    outs = []
    for mod in (a, b):
        f = None
        if with_feats:
            f = torch.from_numpy(rng.standard_normal((sum(n), c)).astype(np.float32)).to(dev) \
                if not outs else outs[0][2].detach().clone()
            f.requires_grad_()
        _, y = mod(xyz, cnt, new_xyz, ncnt, f)
        y.square().sum().backward()
        outs.append((y, mod, f))
    (ya, ma, fa), (yb, mb, fb) = outs
    assert ya.shape == (2 * m_per, 24 + 32)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    if with_feats:
        assert torch.allclose(fa.grad, fb.grad, rtol=1e-3, atol=1e-3 * float(fb.grad.abs().max()))
    for (na, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-3, atol=1e-3 * float(pb.grad.abs().max())), na
    for (na, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-4, atol=1e-5), na


@pytest.mark.parametrize('n,c,relu', [(5000, 16, True), (30336, 32, True), (884736, 64, True), (777, 128, False),
                                      (2, 256, True), (1025, 4, True)])
def test_fused_bn_relu_rows_matches_torch(dev, n, c, relu):
    """dm_bn_rows_forward/backward == relu(F.batch_norm(x)) (float64 reference): outputs, running
    statistics, and the gradients w.r.t. x, gamma, beta."""
    import torch.nn as nn
    from detmatch_amd.bn_relu import bn_relu_rows
    torch.manual_seed(n + c)
    x = (torch.randn(n, c, device=dev) * 2.5 + torch.linspace(-40, 40, c, device=dev)).requires_grad_()
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    y = bn_relu_rows(x, bn, relu=relu)
    g = torch.randn_like(y)
    y.backward(g)
    xd = x.detach().double().requires_grad_()
    pre = ref(xd)
    yd = torch.relu(pre) if relu else pre
    yd.backward(g.double())
    pre = pre.detach()
    scale = float(yd.abs().max()) + 1e-6
    assert float((y.double() - yd).abs().max()) < 2e-5 * scale + 2e-5
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref.running_var, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1
    gs = float(xd.grad.abs().max()) + 1e-9
    # elements whose pre-activation sits within rounding distance of 0 may take the other ReLU branch
    sure = (pre.abs() > 1e-4) if relu else torch.ones_like(pre, dtype=torch.bool)
    assert float(((x.grad.double() - xd.grad).abs() * sure).max()) < 5e-4 * gs
    assert float((~sure).double().mean()) < 1e-3
    assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, rtol=2e-3, atol=2e-3 * float(ref.weight.grad.abs().max()))
    assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, rtol=2e-3, atol=2e-3 * float(ref.bias.grad.abs().max()))
    bn.eval()
    ref.eval()
    ze = ref(x.detach().double())
    ze = torch.relu(ze) if relu else ze
    with torch.no_grad():
        ye = bn_relu_rows(x.detach(), bn, relu=relu)      # inference: dm_bn_rows_eval, one launch
    assert float((ye.double() - ze).abs().max()) < 2e-6 * (float(ze.abs().max()) + 1.0)
    xg = x.detach().clone().requires_grad_()
    yg = bn_relu_rows(xg, bn, relu=relu)                  # evaluation mode under autograd: torch
    assert yg.requires_grad and torch.allclose(yg.double(), ze, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------
# Reference-compiled / reference-held vectors (round 2): HIP against the goldens directly.
def test_iou_and_nms_equal_compiled_reference(dev):
    """tests/golden/iou3d_ref.npz (the reference's iou3d_cpu.cpp compiled here,
    gen_iou3d_golden.py): IoU values within 3e-6 (1-ulp cosine of the heading: the reference's
    host cosf vs the correctly rounded value of the HIP pre-pass), NMS keep lists bit-exact."""
    import os
    from conftest import GOLDEN
    from detmatch_amd import iou3d_nms
    g = np.load(os.path.join(GOLDEN, 'iou3d_ref.npz'))
    ta, tb = torch.from_numpy(g['a']).to(dev), torch.from_numpy(g['b']).to(dev)
    got = iou3d_nms.boxes_iou_bev(ta, tb).cpu().numpy()
    np.testing.assert_allclose(got, g['iou'], rtol=0, atol=3e-6)
    assert (got != g['iou']).mean() < 0.01
    boxes = g['nms_boxes']
    n = len(boxes)
    scores = torch.arange(n, 0, -1, dtype=torch.float32, device=dev)   # already sorted
    for thr in (0.01, 0.1, 0.7, 0.8):
        keep, _ = iou3d_nms.nms_gpu(torch.from_numpy(boxes).to(dev), scores, thr)
        assert np.array_equal(keep.cpu().numpy(), g['keep_%g' % thr]), thr


def test_reference_points_in_boxes_kat_gpu(dev):
    """The reference-held vector of tests/test_models/test_common_modules/test_roiaware_pool3d.py:
    43-71, mapped to the pcdet box convention as in tests/test_oracle_ops.py."""
    from detmatch_amd.roiaware_pool3d import points_in_boxes_gpu
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
    got = points_in_boxes_gpu(torch.from_numpy(pts).to(dev), torch.from_numpy(pc).to(dev))
    assert got.cpu().tolist() == [[0, 0, 0, 0, 0, -1, -1, -1], [-1] * 8]


def test_reference_grouping_kat_gpu(dev):
    """Reference-held grouping_operation vector (test_pointnet_ops.py:126-195), stacked layout."""
    from detmatch_amd import pointnet2_stack as pn
    from test_oracle_ops import reference_grouping_kat
    feats, fc, idx, ic, want = reference_grouping_kat()
    t = lambda a, dt=None: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    got = pn.grouping_operation(t(feats), t(fc, np.int32), t(idx), t(ic, np.int32))
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('n,m,batch', [(60000, 512, 2), (200000, 1024, 1), (30001, 300, 3)])
def test_fps_large_clouds_multi_workgroup_equals_single_workgroup(dev, n, m, batch):
    """Clouds beyond one workgroup's registers: several co-operating workgroups per sample (per-round
    exchange through L2) return the indices of the one-workgroup kernel bit for bit (same distance
    arithmetic, same total order of the tie rule) — incl. ragged samples and duplicate points."""
    from detmatch_amd import _lib, pointnet2_stack as pn2
    g = torch.Generator().manual_seed(n)
    sizes = [n - 17 * b for b in range(batch)]
    pts = [torch.rand(s, 3, generator=g) * torch.tensor([150.0, 150.0, 6.0]) for s in sizes]
    pts[0][1000:1100] = pts[0][2000:2100]            # exact duplicates: ties decided by the index rule
    xyz = torch.cat(pts).to(dev).contiguous()
    cnt = torch.tensor(sizes, dtype=torch.int32)
    out = []
    for variant in (0, 1):
        _lib.lib().dm_fps_set_variant(variant)
        try:
            out.append(pn2.furthest_point_sample_stack(xyz, cnt, m).cpu())
        finally:
            _lib.lib().dm_fps_set_variant(0)
    assert out[0].shape == (batch, m) and torch.equal(out[0], out[1])
    assert all(int(out[0][b].max()) < sizes[b] for b in range(batch))
    assert all(len(set(out[0][b].tolist())) > m * 0.9 for b in range(batch))


def test_exact_bev_overlap_on_near_touching_and_near_threshold_pairs(dev):
    """dm_boxes_overlap_bev_exact (KITTI evaluation, GT-paste collision test) against independent
    float64 polygon clipping (tests/_polyclip.py) on randomized pairs incl. boxes a few millimetres
    apart (must be 0: the NMS kernel's 1 cm corner margin reports an overlap there) and pairs whose IoU
    sits within 1e-3 of the 0.7 / 0.5 / 0.25 thresholds (the TP / FP decision must agree)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from _polyclip import intersection_area
    from detmatch_amd import iou3d_nms
    rng = np.random.default_rng(3)
    n = 400
    a = np.zeros((n, 7), np.float32)
    a[:, 0:2] = rng.uniform(-20, 20, (n, 2))
    a[:, 3:5] = rng.uniform(1.0, 5.0, (n, 2))
    a[:, 5] = 1
    a[:, 6] = rng.uniform(-3.2, 3.2, n)
    b = a.copy()
    b[:, 0:2] += rng.normal(0, 1.0, (n, 2))
    b[:, 3:5] *= rng.uniform(0.8, 1.25, (n, 2))
    b[:, 6] += rng.normal(0, 0.3, n)
    # near-touching: same heading, shifted along the local x axis by (dx_a + dx_b) / 2 + gap
    for i, gap in enumerate([0.002, 0.005, 0.009, -0.002, 0.0005, 0.02]):
        b[i] = a[i]
        h = a[i, 6]
        d = a[i, 3] + gap
        b[i, 0] += d * np.cos(h)
        b[i, 1] += d * np.sin(h)

    def corners(r):      # counter-clockwise turn by the heading: the pcdet kernel's convention
        c, s = np.cos(r[6]), np.sin(r[6])
        loc = np.array([[-r[3] / 2, -r[4] / 2], [r[3] / 2, -r[4] / 2], [r[3] / 2, r[4] / 2], [-r[3] / 2, r[4] / 2]])
        return np.stack([r[0] + c * loc[:, 0] - s * loc[:, 1], r[1] + s * loc[:, 0] + c * loc[:, 1]], 1)
    want = np.array([intersection_area(corners(a[i].astype(np.float64)), corners(b[i].astype(np.float64)))
                     for i in range(n)])
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    got = iou3d_nms.boxes_overlap_bev_exact(ta, tb).diagonal().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    assert (got[[0, 1, 2, 4, 5]] == 0).all() and got[3] > 0          # gaps of 0.5-20 mm: disjoint
    margin = iou3d_nms.boxes_overlap_bev(ta, tb).diagonal().cpu().numpy()
    assert (margin[[0, 1, 2]] > 0).all()                             # what the corner margin does to them
    # near-threshold IoUs: scale b around the pair's centre until IoU is within 1e-3 of a threshold
    area_a, area_b = a[:, 3] * a[:, 4], b[:, 3] * b[:, 4]
    iou_w = want / (area_a + area_b - want)
    iou_g = got / (area_a + area_b - got)
    for thr in (0.7, 0.5, 0.25):
        close = np.abs(iou_w - thr) > 1e-6
        assert ((iou_w > thr) == (iou_g > thr))[close].all()


def test_anchor_decode_kernel_is_the_tensor_chain(dev):
    """dm_anchor_decode == AnchorHeadTemplate.generate_predicted_boxes' tensor chain (ResidualCoder
    decode + direction-bin correction), bit for bit, on the KITTI anchor grid."""
    from detmatch_amd import configs, fused
    from detmatch_amd.pcdet.config import ConfigDict
    from detmatch_amd.pcdet.dense_heads import AnchorHeadSingle
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    head = AnchorHeadSingle(cfg.DENSE_HEAD, input_channels=16, num_class=3, class_names=configs.CLASS_NAMES,
                            grid_size=np.array([1408, 1600, 40]),
                            point_cloud_range=np.array(configs.POINT_CLOUD_RANGE, dtype=np.float32)).to(dev).eval()
    head.boxes_detached_downstream = True      # what PVRCNN.build_networks declares
    anchors = head._cat_anchors()
    n = anchors.view(-1, 7).shape[0]
    g = torch.Generator().manual_seed(5)
    box = (torch.randn(2, n * 7, generator=g) * 0.7).to(dev).view(2, -1, 7 * 6)
    cls = torch.randn(2, n * 3, generator=g).to(dev).view(2, -1, 3 * 6)
    dirs = torch.randn(2, n * 2, generator=g).to(dev).view(2, -1, 2 * 6)
    dirs.view(-1, 2)[::7] = 0.25                       # ties between the two direction bins
    assert head.boxes_detached_downstream
    with torch.no_grad():
        _, a = head.generate_predicted_boxes(2, cls, box, dirs)
        prev = fused.ENABLED
        fused.ENABLED = False
        try:
            _, b = head.generate_predicted_boxes(2, cls, box, dirs)
        finally:
            fused.ENABLED = prev
    assert a.shape == b.shape == (2, n, 7)
    assert torch.equal(a, b)


@pytest.mark.parametrize('rows,k,n', [(70000, 132, 64), (65537, 64, 64), (40000, 64, 128), (33000, 64, 132),
                                      (9000, 20, 16), (64, 8, 4), (100000, 36, 32)])
def test_rowgemm_matches_float64_matmul(dev, rows, k, n):
    """csrc/rowgemm.hip (shared-MLP GEMM with the weights resident in LDS) == x @ w^T, and the strided
    form used for the input gradient of a first layer: dead leading columns are zeros."""
    from detmatch_amd import _lib
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    want = x.double() @ w.double().t()
    L = _lib.lib()
    assert L.dm_rowgemm_supported(k, n)
    y = torch.full((rows, n), float('nan'), device=dev)
    _lib.check(L.dm_rowgemm(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), rows, k, n, _lib.stream()), 'dm_rowgemm')
    scale = float(want.abs().max())
    assert float((y.double() - want).abs().max()) <= 2e-6 * scale * k ** 0.5
    if n <= 128:
        y2 = torch.full((rows, n + 8), float('nan'), device=dev)
        _lib.check(L.dm_rowgemm_strided(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y2), rows, k, n, n + 8, 8,
                                        _lib.stream()), 'dm_rowgemm_strided')
        assert float(y2[:, :8].abs().max()) == 0.0
        assert torch.equal(y2[:, 8:], y)
        # the transposed-weight entry (input gradient from the stored weight): wt (k, 4 + n) holds w^T behind 4 columns
        wt = torch.randn(k, 4 + n, generator=g).to(dev)
        wt[:, 4:] = w.t()
        y3 = torch.full((rows, n + 8), float('nan'), device=dev)
        _lib.check(L.dm_rowgemm_wt(_lib.ptr(x), wt.data_ptr() + 16, 4 + n, _lib.ptr(y3), rows, k, n, n + 8, 8,
                                   _lib.stream()), 'dm_rowgemm_wt')
        assert torch.equal(y3, y2)


@pytest.mark.parametrize('rows,k,n', [(70000, 132, 64), (65537, 64, 64), (40000, 36, 32), (33000, 16, 16),
                                      (9000, 20, 16), (4100, 160, 64), (131072, 68, 64), (5000, 100, 48)])
def test_tall_wgrad_matches_float64(dev, rows, k, n):
    """dm_tall_wgrad (dW = dY^T X over 10^4..10^6 rows, csrc/conv2d.hip: split arithmetic, streaming row ranges)
    against float64: maximum error within 4 fp32 ulps of the natural scale sum |dy||x| per sqrt(rows) accumulated
    products (the bound of the dense convolutions' split arithmetic), bitwise reproducible, and `accumulate`."""
    from detmatch_amd import _lib
    g = torch.Generator().manual_seed(rows + 3 * k + n)
    x = torch.randn(rows, k, generator=g).to(dev)
    gy = torch.randn(rows, n, generator=g).to(dev)
    L = _lib.lib()
    assert L.dm_tall_wgrad_supported(n, k)
    ws = torch.empty(int(L.dm_tall_wgrad_workspace_bytes(rows, n, k)), dtype=torch.uint8, device=dev)

    def run(dst, acc):
        _lib.check(L.dm_tall_wgrad(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(dst), rows, n, k, acc, _lib.ptr(ws), ws.numel(),
                                   _lib.stream()), 'dm_tall_wgrad')
    a = torch.full((n, k), float('nan'), device=dev)
    b = torch.full((n, k), float('nan'), device=dev)
    run(a, 0)
    run(b, 0)
    assert torch.equal(a, b)
    want = gy.double().t() @ x.double()
    scale = gy.double().abs().t() @ x.double().abs()
    err = float(((a.double() - want).abs() / scale).max())
    assert err <= 4 * 2.0 ** -24 * rows ** 0.5, err
    run(b, 1)
    torch.testing.assert_close(b, 2 * a, rtol=1e-6, atol=1e-6)


def test_tall_skinny_linear_autograd_through_rowgemm(dev):
    """TallSkinnyLinear on the rowgemm path (threshold lowered) == the BLAS path: output, weight gradient,
    input gradient of the live columns; the dead leading columns get zeros."""
    from detmatch_amd.pointnet2_stack import TallSkinnyLinear
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(50000, 68, generator=g).to(dev)
    w0 = (torch.randn(64, 68, generator=g) / 8).to(dev)
    gy = torch.randn(50000, 64, generator=g).to(dev)
    res = []
    prev = TallSkinnyLinear.ROWGEMM_MIN_ROWS
    try:
        for thr in (0, 10 ** 9):
            TallSkinnyLinear.ROWGEMM_MIN_ROWS = thr
            x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            y = TallSkinnyLinear.apply(x, w, 4)
            y.backward(gy)
            res.append((y.detach(), x.grad.clone(), w.grad.clone()))
    finally:
        TallSkinnyLinear.ROWGEMM_MIN_ROWS = prev
    (ya, xa, wa), (yb, xb, wb) = res
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
    assert float(xa[:, :4].abs().max()) == 0.0
    assert torch.allclose(xa[:, 4:], xb[:, 4:], rtol=1e-4, atol=1e-4)
    assert torch.allclose(wa, wb, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('m,ns,c', [(4096, 16, 32), (1000, 32, 64), (55296, 16, 64), (7, 5, 128), (300, 16, 16)])
def test_fused_bn_relu_maxpool_matches_two_step(dev, m, ns, c):
    """bn_relu_rows_max (BatchNorm + ReLU + max over nsample in one apply pass, backward from the pooled
    gradient) == relu(bn(x)).view(M, ns, C).max(1) through the row kernels + torch.max: pooled values bit
    for bit, running statistics equal, gradients w.r.t. x / gamma / beta equal."""
    import torch.nn as nn
    from detmatch_amd.bn_relu import bn_relu_rows, bn_relu_rows_max
    torch.manual_seed(m + c)
    x0 = torch.randn(m * ns, c, device=dev) * 1.7 + torch.linspace(-3, 3, c, device=dev)
    gp = torch.randn(m, c, device=dev)
    res = []
    for fused in (True, False):
        bn = nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).to(dev)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
        x = x0.clone().requires_grad_(True)
        if fused:
            y = bn_relu_rows_max(x, bn, ns)
            assert 'BNReLUMaxRows' in type(y.grad_fn).__name__
        else:
            y = bn_relu_rows(x, bn, relu=True).view(m, ns, c).max(dim=1)[0]
        y.backward(gp)
        res.append((y.detach(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    a, b = res
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]) and a[6] == b[6] == 1
    gs = float(b[1].abs().max()) + 1e-12
    assert float((a[1] - b[1]).abs().max()) <= 2e-5 * gs
    for u, v in ((a[2], b[2]), (a[3], b[3])):
        assert float((u - v).abs().max()) <= 1e-4 * (float(v.abs().max()) + 1e-6)
    # evaluation mode (the EMA teacher): one launch, equal to the two-step form
    bn.eval()
    with torch.no_grad():
        ye = bn_relu_rows_max(x0, bn, ns)
        want = bn_relu_rows(x0, bn, relu=True).view(m, ns, c).max(dim=1)[0]
    assert torch.equal(ye, want)
    # the pooled gradient as a column block of a wider matrix (the gradient of concatenated grouper outputs): read
    # in place through its row pitch, bit-identical to the contiguous copy
    wide = torch.randn(m, c + 36, device=dev)
    wide[:, 4:4 + c] = gp
    outs = []
    for g_in in (gp, wide[:, 4:4 + c]):
        bn2 = nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).to(dev)
        x = x0.clone().requires_grad_(True)
        bn_relu_rows_max(x, bn2, ns).backward(g_in)
        outs.append((x.grad.clone(), bn2.weight.grad.clone(), bn2.bias.grad.clone()))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.parametrize('counts', [([5000, 2300], [2048, 2048]), ([3000, 0, 1700], [253, 0, 300]),
                                    ([700, 64], [9001, 8999])])
def test_ball_query_pair_equals_two_queries(orc, dev, counts):
    """dm_ball_query_stack2 (two radii in one scan) == the oracle's ball query run once per radius:
    indices and empty flags bit for bit, on both launch shapes."""
    from detmatch_amd import pointnet2_stack as pn
    xyz_cnt, new_cnt = counts
    rng = np.random.default_rng(sum(new_cnt))
    xyz = _stacked(rng, xyz_cnt, 0, 6)
    new_xyz = _stacked(rng, new_cnt, -0.5, 6.5)
    t = lambda a, dt=None: torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)
    for (ra, na), (rb, nb) in (((0.4, 16), (0.8, 16)), ((1.2, 32), (0.3, 16)), ((2.4, 16), (4.8, 32))):
        (ia, ea), (ib, eb) = pn.ball_query_pair(ra, na, rb, nb, t(xyz), t(xyz_cnt, np.int32), t(new_xyz),
                                                t(new_cnt, np.int32))
        for (idx, empty), (r, n) in (((ia, ea), (ra, na)), ((ib, eb), (rb, nb))):
            oidx, oempty = orc.ball_query(r, n, xyz, xyz_cnt, new_xyz, new_cnt)
            assert np.array_equal(idx.cpu().numpy(), oidx)
            assert np.array_equal(empty.cpu().numpy(), oempty)


@pytest.mark.parametrize('rows,k,n,ns', [(70016, 132, 64, 16), (65536, 36, 32, 16), (131072, 20, 16, 32),
                                         (40000, 64, 128, 16), (33007 * 16, 64, 64, 16)])
def test_rowgemm_statistics_feed_the_batchnorm(dev, rows, k, n, ns):
    """dm_rowgemm_stats: the column statistics reduced in the GEMM's epilogue give the same BatchNorm
    (+ReLU, + max over nsample) output, running statistics and gradients as the separate statistics pass."""
    import torch.nn as nn
    from detmatch_amd.bn_relu import bn_relu_rows, bn_relu_rows_max
    from detmatch_amd.pointnet2_stack import TallSkinnyLinear
    g = torch.Generator().manual_seed(rows % 1000 + k)
    x0 = (torch.randn(rows, k, generator=g) + 0.3).to(dev)
    w0 = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    prev = TallSkinnyLinear.ROWGEMM_MIN_ROWS
    TallSkinnyLinear.ROWGEMM_MIN_ROWS = 0
    try:
        for pooled in (False, True):
            res = []
            for stats in (True, False):
                bn = nn.BatchNorm2d(n, eps=1e-3, momentum=0.01).to(dev)
                x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
                y = TallSkinnyLinear.apply(x, w, 0, stats)
                assert (getattr(y, 'dm_bn_pre', None) is not None) == stats
                out = bn_relu_rows_max(y, bn, ns) if pooled else bn_relu_rows(y, bn, relu=True)
                out.backward(torch.ones_like(out) * 0.5 + out.detach() * 0.1)
                res.append((out.detach(), bn.running_mean.clone(), bn.running_var.clone(), x.grad.clone(),
                            bn.weight.grad.clone()))
            a, b = res
            assert float((a[0] - b[0]).abs().max()) <= 2e-5 * (float(b[0].abs().max()) + 1.0)
            assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-7) and torch.allclose(a[2], b[2], rtol=1e-4, atol=1e-7)
            assert float((a[3] - b[3]).abs().max()) <= 1e-4 * (float(b[3].abs().max()) + 1e-9)
            assert float((a[4] - b[4]).abs().max()) <= 1e-3 * (float(b[4].abs().max()) + 1e-6)
    finally:
        TallSkinnyLinear.ROWGEMM_MIN_ROWS = prev


@pytest.mark.parametrize('c', [32, 128])
def test_query_group_rows_half_wave_kernel(dev, c):
    """The 16-references-per-wave grouping kernel (taken from 64 k references) == rows assembled with
    tensor indexing from the ball-query result: [xyz - centre, 0, features], zero rows for empty balls."""
    from detmatch_amd import pointnet2_stack as pn
    rng = np.random.default_rng(c)
    xyz_cnt, new_cnt = [3000, 1700], [2600, 2400]
    xyz = torch.from_numpy(_stacked(rng, xyz_cnt, 0, 12)).to(dev)
    new_xyz = torch.from_numpy(_stacked(rng, new_cnt, -2, 14)).to(dev)          # some centres see nothing
    feats = torch.from_numpy(rng.standard_normal((sum(xyz_cnt), c)).astype(np.float32)).to(dev)
    xc = torch.tensor(xyz_cnt, dtype=torch.int32, device=dev)
    nc = torch.tensor(new_cnt, dtype=torch.int32, device=dev)
    ns = 16
    rows, idx = pn.query_group_rows(0.9, ns, xyz, xc, new_xyz, nc, feats, True)
    assert rows.shape == (5000, ns, 4 + c) and 5000 * ns >= 65536
    _, empty = pn.ball_query(0.9, ns, xyz, xc, new_xyz, nc)
    assert 0 < int(empty.sum()) < 5000
    start = torch.cat([torch.zeros(new_cnt[0], dtype=torch.long), torch.full((new_cnt[1],), xyz_cnt[0])]).to(dev)
    src = idx.long() + start[:, None]
    want = torch.cat([xyz[src] - new_xyz[:, None, :], torch.zeros(5000, ns, 1, device=dev), feats[src]], dim=2)
    want[empty] = 0
    assert torch.equal(rows, want)


@pytest.mark.parametrize('kind', ['rot', 'normal', '2d'])
def test_batched_nms_equals_per_sample_calls(dev, kind):
    """dm_nms_batch / dm_nms_2d_batch: B problems in one launch chain give the keep lists of B single calls (both
    phases: the survivor budget is met inside the leading blocks for one sample and not for another)."""
    from detmatch_amd import _lib
    L = _lib.lib()
    torch.manual_seed(3)
    bsz, n, post = 3, 1500, 100
    if kind == '2d':
        xy = torch.rand(bsz, n, 2, device=dev) * 300
        wh = torch.rand(bsz, n, 2, device=dev) * 40 + 4
        boxes = torch.cat([xy, xy + wh], dim=2)
        boxes[1, :, :2] = boxes[1, :1, :2]            # one sample of near-duplicates: few survivors, second phase runs
        boxes[1, :, 2:] = boxes[1, :, :2] + 30 + torch.rand(n, 2, device=dev)
    else:
        boxes = torch.rand(bsz, n, 7, device=dev)
        boxes[:, :, :2] *= 60
        boxes[:, :, 3:6] = boxes[:, :, 3:6] * 3 + 1
        boxes[:, :, 6] = boxes[:, :, 6] * 6.28
        boxes[1, :, :3] = boxes[1, :1, :3] + torch.rand(n, 3, device=dev) * 0.3
    boxes = boxes.contiguous()
    ws1 = torch.empty((L.dm_nms_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    wsb = torch.empty((L.dm_nms_workspace_bytes(n) * bsz,), dtype=torch.uint8, device=dev)
    keep1 = torch.full((bsz, n), -1, dtype=torch.int64, device=dev)
    num1 = torch.zeros((bsz,), dtype=torch.int32, device=dev)
    for b in range(bsz):
        if kind == '2d':
            rc = L.dm_nms_2d(_lib.ptr(boxes[b]), n, 0.5, post, _lib.ptr(keep1[b]), _lib.ptr(num1[b:b + 1]), _lib.ptr(ws1),
                             ws1.numel(), _lib.stream())
        else:
            fn = L.dm_nms if kind == 'rot' else L.dm_nms_normal
            rc = fn(_lib.ptr(boxes[b]), n, 0.5, post, _lib.ptr(keep1[b]), _lib.ptr(num1[b:b + 1]), _lib.ptr(ws1), ws1.numel(),
                    _lib.stream())
        _lib.check(rc, 'single')
    keepb = torch.full((bsz, n), -1, dtype=torch.int64, device=dev)
    numb = torch.zeros((bsz,), dtype=torch.int32, device=dev)
    if kind == '2d':
        rc = L.dm_nms_2d_batch(_lib.ptr(boxes), bsz, n, 0.5, post, _lib.ptr(keepb), keepb.stride(0), _lib.ptr(numb),
                               _lib.ptr(wsb), wsb.numel(), _lib.stream())
    else:
        rc = L.dm_nms_batch(_lib.ptr(boxes), bsz, n, 0.5, post, 0 if kind == 'rot' else 1, _lib.ptr(keepb), keepb.stride(0),
                            _lib.ptr(numb), _lib.ptr(wsb), wsb.numel(), _lib.stream())
    _lib.check(rc, 'batch')
    assert torch.equal(num1, numb)
    assert int(num1.min()) < post <= int(num1.max()) or int(num1.max()) == post
    for b in range(bsz):
        k = int(num1[b])
        assert torch.equal(keep1[b, :k], keepb[b, :k])


def test_voxel_centers_and_counts_kernel(dev):
    """dm_voxel_centers: centres bit-identical to get_voxel_centers' tensor chain, rows per sample equal to the
    counting formulation — ragged samples, an empty sample in the middle, strided level geometry."""
    from detmatch_amd.pcdet import pfe
    from detmatch_amd.pcdet.utils import get_voxel_centers
    rng = np.random.default_rng(4)
    sizes = [700, 0, 1301, 5]
    coords = np.concatenate([np.concatenate([np.full((n, 1), b), rng.integers(0, [11, 400, 352], (n, 3))], 1)
                             for b, n in enumerate(sizes)]).astype(np.int32)
    c = torch.from_numpy(coords).to(dev)
    for down in (1, 4):
        xyz, cnt = pfe.voxel_centers_and_counts(c, len(sizes), down, [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1])
        want = get_voxel_centers(c[:, 1:4], downsample_times=down, voxel_size=[0.05, 0.05, 0.1],
                                 point_cloud_range=[0, -40, -3, 70.4, 40, 1])
        assert torch.equal(xyz, want.contiguous())
        assert cnt.dtype == torch.int32 and cnt.cpu().tolist() == sizes
    assert torch.equal(cnt, pfe.batch_row_counts(c[:, 0], len(sizes)))
