"""oracle/mmdet_ref.py (independent per-image restatement of the mmdet 2.14 / mmcv 1.3.16 algorithms,
test infrastructure) against the product's HOST formulation of the 2D branch (mm2d/faster_rcnn.py dense
tensor path, mm3d/losses.py costs / losses) on the CPU.  The GPU counterpart (the fused kernels against
the same checker) is tests/test_mmdet_parity_gpu.py."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs

SIZES = [(12, 40), (6, 20), (3, 10), (2, 5), (1, 3)]
STRIDES = [4, 8, 16, 32, 64]


@pytest.fixture(scope='module')
def ref():
    from oracle import mmdet_ref
    return mmdet_ref


def _boxes(rng, n, w, h, lo=8, hi=60):
    c = rng.uniform([0, 0], [w, h], size=(n, 2))
    s = rng.uniform(lo, hi, size=(n, 2))
    return np.concatenate([np.clip(c - s / 2, 0, None), np.minimum(c + s / 2, [w, h])], 1).astype(np.float32)


def _frcnn():
    from detmatch_amd.mm2d.faster_rcnn import FasterRCNN
    cfg = configs.frcnn_kitti_model()
    cfg.pop('type')
    torch.manual_seed(0)
    return FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg)


def test_anchors_and_coder(ref):
    from detmatch_amd.mm2d.faster_rcnn import AnchorGenerator, DeltaXYWHBBoxCoder
    ag = AnchorGenerator(strides=STRIDES, ratios=[0.5, 1.0, 2.0], scales=[8])
    got = ag.grid_anchors(SIZES, 'cpu')
    want = ref.grid_anchors(SIZES, STRIDES, [8], [0.5, 1.0, 2.0])
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.numpy(), w, rtol=0, atol=1e-4)
    rng = np.random.default_rng(0)
    a, b = _boxes(rng, 50, 300, 100), _boxes(rng, 50, 300, 100)
    coder = DeltaXYWHBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2))
    enc = coder.encode(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    np.testing.assert_allclose(enc, ref.bbox2delta(a, b, stds=(0.1, 0.1, 0.2, 0.2)), rtol=1e-5, atol=1e-5)
    d = rng.normal(0, 2, size=(50, 12)).astype(np.float32)
    dec = coder.decode(torch.from_numpy(a), torch.from_numpy(d), max_shape=(100, 300)).numpy()
    np.testing.assert_allclose(dec, ref.delta2bbox(a, d, stds=(0.1, 0.1, 0.2, 0.2), max_shape=(100, 300)),
                               rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('low', [True, False])
def test_max_iou_assign(ref, low):
    from detmatch_amd.mm2d.faster_rcnn import max_iou_assign
    rng = np.random.default_rng(1)
    boxes = _boxes(rng, 500, 200, 120, 10, 70)
    gt = boxes[rng.permutation(500)[:9]] + 2
    gt[3] = gt[2]                                       # duplicate GT: ties, the later one wins
    got = max_iou_assign(torch.from_numpy(boxes), torch.from_numpy(gt), 0.7, 0.3, 0.3, low).numpy()
    want, _ = ref.max_iou_assign(boxes, gt, 0.7, 0.3, 0.3, low)
    assert np.array_equal(got, want)
    assert np.array_equal(max_iou_assign(torch.from_numpy(boxes), torch.from_numpy(gt[:0]), 0.7, 0.3, 0.3, low).numpy(),
                          ref.max_iou_assign(boxes, gt[:0], 0.7, 0.3, 0.3, low)[0])


@pytest.mark.parametrize('n_gt', [(5, 2), (0, 3), (0, 0)])
def test_rpn_loss_host_path(ref, n_gt):
    m = _frcnn()
    head = m.rpn_head
    with torch.no_grad():
        head.rpn_cls.weight.normal_(std=0.05)
        head.rpn_reg.weight.normal_(std=0.02)
    rng = np.random.default_rng(3)
    gts = [_boxes(rng, k, 160, 48, 12, 60) for k in n_gt]
    feats = [torch.randn(2, 256, h, w, generator=torch.Generator().manual_seed(i)) for i, (h, w) in enumerate(SIZES)]
    cls, reg = head(feats)
    n_anchor = sum(h * w * 3 for h, w in SIZES)
    keys = torch.rand((2, n_anchor), generator=torch.Generator().manual_seed(7))
    cls_l = [c.detach().clone().requires_grad_(True) for c in cls]
    reg_l = [r.detach().clone().requires_grad_(True) for r in reg]
    got = head.loss(cls_l, reg_l, [torch.from_numpy(g) for g in gts], None, fused=False, keys=keys)
    (got['loss_rpn_cls'] * 0.7 + got['loss_rpn_bbox'] * 1.3).backward()
    anchors = np.concatenate(ref.grid_anchors(SIZES, STRIDES, [8], [0.5, 1.0, 2.0]))
    cfg = configs.frcnn_train_cfg()['rpn']
    cls_r = [c.detach().double().requires_grad_(True) for c in cls]
    reg_r = [r.detach().double().requires_grad_(True) for r in reg]
    lc, lb, targets = ref.rpn_loss(cls_r, reg_r, anchors, gts, cfg['assigner'], cfg['sampler'], keys.numpy())
    (lc * 0.7 + lb * 1.3).backward()
    assert float(got['loss_rpn_cls']) == pytest.approx(float(lc), rel=1e-5)
    assert float(got['loss_rpn_bbox']) == pytest.approx(float(lb), rel=1e-5, abs=1e-9)
    for a, b in zip(cls_l + reg_l, cls_r + reg_r):
        np.testing.assert_allclose(a.grad.numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-8)
    if n_gt[0] == 0:            # an image without positives still counts one in num_total_samples
        n_neg = [len(t[5]) for t in targets]
        assert n_neg[0] == 256


def _proposals(rng, gts, n=300):
    out = []
    for gt in gts:
        p = _boxes(rng, n, 160, 48, 6, 50)
        if len(gt):
            src = rng.integers(0, len(gt), size=n // 2)
            p[: n // 2] = gt[src] + rng.normal(0, 2.5, size=(n // 2, 4)).astype(np.float32)
        ok = rng.uniform(size=n) < 0.85
        out.append((np.concatenate([p, rng.uniform(size=(n, 1)).astype(np.float32)], 1), ok))
    return out


def sort_rows(*cols):
    """Rows of the column-stacked arrays in lexicographic order (set comparison of sampled RoIs)."""
    m = np.concatenate([np.asarray(c, np.float64).reshape(len(cols[0]), -1) for c in cols], 1)
    return m[np.lexsort(m.T[::-1])]


def roi_target_rows(ref, props, gts, gls, keys, num_classes=3):
    """The checker's sampled RoI targets of a batch in the product's row format [b, roi, label, weight,
    targets, box weights], valid proposals compacted as mmdet sees them (variable length)."""
    cfg = configs.frcnn_train_cfg()['rcnn']
    rows = []
    for i, ((p, ok), gt, gl) in enumerate(zip(props, gts, gls)):
        k = len(gt)
        kk = np.concatenate([keys[i][:k], keys[i][k:k + len(p)][ok]]) if k else keys[i][:len(p)][ok]
        rois, labels, lw, bt, bw = ref.roi_targets_single(p[ok], gt, gl, cfg['assigner'], cfg['sampler'], kk,
                                                          num_classes)
        rows.append(np.concatenate([np.full((len(rois), 1), i), rois, labels[:, None], lw[:, None], bt, bw], 1))
    return np.concatenate(rows)


@pytest.mark.parametrize('n_gt', [(6, 3), (0, 2), (0, 0)])
def test_roi_targets_host_path(ref, n_gt):
    m = _frcnn()
    rng = np.random.default_rng(9)
    gts = [_boxes(rng, k, 160, 48, 12, 60) for k in n_gt]
    gls = [rng.integers(0, 3, size=k) for k in n_gt]
    props = _proposals(rng, gts)
    keys = rng.uniform(size=(2, 300 + max(n_gt))).astype(np.float32)
    got = m.roi_head._targets_tensor([(torch.from_numpy(p), torch.from_numpy(ok)) for p, ok in props],
                                     [torch.from_numpy(g) for g in gts], [torch.from_numpy(g) for g in gls],
                                     torch.from_numpy(keys))
    rois, labels, lw, tg, bw = [t.numpy() for t in got]
    live = lw > 0
    want = roi_target_rows(ref, props, gts, gls, keys)
    have = np.concatenate([rois, labels[:, None], lw[:, None], tg, bw], 1)[live]
    assert have.shape == want.shape
    np.testing.assert_allclose(sort_rows(have), sort_rows(want), rtol=1e-5, atol=1e-5)


def test_bbox_head_loss_host_path(ref):
    m = _frcnn()
    bh = m.roi_head.bbox_head
    g = torch.Generator().manual_seed(4)
    n = 300
    labels = torch.randint(0, 4, (n,), generator=g)
    lw = (torch.rand(n, generator=g) < 0.8).float()
    tg = torch.randn(n, 4, generator=g)
    bw = (labels < 3).float()[:, None].expand(-1, 4).contiguous()
    cs = (torch.randn(n, 4, generator=g) * 2).requires_grad_(True)
    bp = torch.randn(n, 12, generator=g).requires_grad_(True)
    got = bh.loss(cs, bp, labels, lw, tg, bw, fused=False)
    (got['loss_cls'] * 1.3 + got['loss_bbox'] * 0.4).backward()
    keep = lw > 0                                   # mmdet's batch holds the sampled rows only
    cr, br = cs.detach()[keep].double().requires_grad_(True), bp.detach()[keep].double().requires_grad_(True)
    lc, lb, acc = ref.bbox_head_loss(cr, br, labels[keep], lw[keep], tg[keep], bw[keep], 3, alpha=0.5)
    (lc * 1.3 + lb * 0.4).backward()
    assert float(got['loss_cls']) == pytest.approx(float(lc), rel=1e-5)
    assert float(got['loss_bbox']) == pytest.approx(float(lb), rel=1e-5)
    assert float(got['acc']) == pytest.approx(float(acc), rel=1e-5)
    np.testing.assert_allclose(cs.grad[keep].numpy(), cr.grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(bp.grad[keep].numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-9)
    assert float(cs.grad[~keep].abs().max()) == 0.0


def test_match_costs_and_consistency_losses(ref):
    from detmatch_amd.mm3d import losses as L
    rng = np.random.default_rng(5)
    f = np.array([1242, 375, 1242, 375], np.float32)
    pred, gt = _boxes(rng, 30, 1242, 375, 10, 200), _boxes(rng, 11, 1242, 375, 10, 200)
    pn = ref.bbox_xyxy_to_cxcywh(pred) / f
    logits = rng.normal(0, 2, size=(30, 3)).astype(np.float32)
    gl = rng.integers(0, 3, size=11)
    np.testing.assert_allclose(L.FocalLossCost(weight=2.0)(torch.from_numpy(logits), torch.from_numpy(gl)).numpy(),
                               ref.focal_loss_cost(logits, gl, 2.0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(L.BBoxL1Cost(weight=5.0)(torch.from_numpy(pn), torch.from_numpy(gt / f)).numpy(),
                               ref.bbox_l1_cost(pn, gt / f, 5.0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(L.IoUCost(iou_mode='giou', weight=2.0)(torch.from_numpy(pred), torch.from_numpy(gt)).numpy(),
                               ref.iou_cost(pred, gt, 2.0), rtol=1e-5, atol=1e-5)
    # consistency losses (consumers_3d.py:84-99 builds mmdet FocalLoss / L1Loss / GIoULoss)
    a, b = torch.from_numpy(pred[:11]), torch.from_numpy(gt)
    assert float(L.GIoULoss()(a, b)) == pytest.approx(float(ref.giou_loss_mean(a, b)), rel=1e-5)
    assert float(L.L1Loss()(a / 1242, b / 1242)) == pytest.approx(float(ref.l1_loss_mean(a / 1242, b / 1242)), rel=1e-5)
    lg = torch.from_numpy(logits[:11])
    assert float(L.FocalLoss(alpha=0.25)(lg, torch.from_numpy(gl))) == pytest.approx(
        float(ref.focal_loss_mean(lg, torch.from_numpy(gl), alpha=0.25)), rel=1e-5)


def test_nms_and_roi_levels(ref):
    rng = np.random.default_rng(6)
    boxes = _boxes(rng, 400, 300, 200, 10, 80)
    scores = rng.uniform(size=400).astype(np.float32)
    keep = ref.nms(boxes, scores, 0.5)
    assert len(keep) == len(set(keep.tolist())) and np.all(np.diff(scores[keep]) <= 0)
    ov = ref.bbox_overlaps(boxes[keep], boxes[keep])
    np.fill_diagonal(ov, 0)
    assert float(ov.max()) <= 0.5                       # survivors do not overlap beyond the threshold
    dropped = np.setdiff1d(np.arange(400), keep)
    ovd = ref.bbox_overlaps(boxes[dropped], boxes[keep])
    better = scores[keep][None, :] >= scores[dropped][:, None]
    assert bool(((ovd > 0.5) & better).any(1).all())    # every dropped box has a better-scored suppressor
    dets, k2 = ref.batched_nms(boxes, scores, rng.integers(0, 3, size=400), 0.5)
    assert len(k2) >= len(keep) and dets.shape[1] == 5
    rois = np.concatenate([np.zeros((5, 1), np.float32),
                           np.array([[0, 0, 10, 10], [0, 0, 111, 111], [0, 0, 112, 112], [0, 0, 300, 300], [0, 0, 900, 900]], np.float32)], 1)
    assert ref.map_roi_levels(rois, 4).tolist() == [0, 0, 1, 2, 3]
