"""Row G on the GPU: the fused assignment / sampling / loss / proposal kernels of the 2D detector
(csrc/det2d_targets.hip, through the C-ABI) against oracle/mmdet_ref.py — the independent per-image
restatement of mmdet 2.14's MaxIoUAssigner, RandomSampler (same keys), DeltaXYWHBBoxCoder, AnchorHead.loss,
RPNHead._get_bboxes_single + mmcv batched_nms, BBoxHead.get_targets / loss — NOT against the product's own
tensor path.  Sampled sets / labels / kept proposals exact, losses 1e-4, gradients 1e-3 of their scale."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs
from test_mmdet_ref import STRIDES, _boxes, _frcnn, _proposals, roi_target_rows, sort_rows

pytestmark = pytest.mark.gpu

FULL = [(96, 312), (48, 156), (24, 78), (12, 39), (6, 20)]         # 384 x 1248 image
SMALL = [(24, 80), (12, 40), (6, 20), (3, 10), (2, 5)]


@pytest.fixture(scope='module')
def ref():
    from oracle import mmdet_ref
    return mmdet_ref


def _head(dev):
    m = _frcnn().to(dev)
    with torch.no_grad():
        m.rpn_head.rpn_cls.weight.normal_(std=0.05)
        m.rpn_head.rpn_reg.weight.normal_(std=0.02)
    return m


@pytest.mark.parametrize('sizes,n_gt', [(SMALL, (5, 2)), (SMALL, (0, 3)), (SMALL, (0, 0)), (FULL, (9, 4))])
def test_rpn_loss_kernel_against_mmdet_checker(dev, ref, sizes, n_gt):
    m = _head(dev)
    head = m.rpn_head
    w, h = sizes[0][1] * 4, sizes[0][0] * 4
    rng = np.random.default_rng(3)
    gts = [_boxes(rng, k, w, h, 12, 0.4 * h) for k in n_gt]
    feats = [torch.randn(2, 256, fh, fw, generator=torch.Generator().manual_seed(i)).to(dev)
             .contiguous(memory_format=torch.channels_last) for i, (fh, fw) in enumerate(sizes)]
    cls, reg = head(feats)
    for y in head._raw_levels:
        y.retain_grad()
    n_anchor = sum(fh * fw * 3 for fh, fw in sizes)
    keys = torch.rand((2, n_anchor), generator=torch.Generator().manual_seed(7))
    got = head.loss(cls, reg, [torch.from_numpy(g).to(dev) for g in gts], None, keys=keys.to(dev))
    assert 'FusedRpnLoss' in type(got['loss_rpn_cls'].grad_fn.next_functions[0][0]).__name__
    (got['loss_rpn_cls'] * 0.7 + got['loss_rpn_bbox'] * 1.3).backward()
    anchors = np.concatenate(ref.grid_anchors(sizes, STRIDES, [8], [0.5, 1.0, 2.0]))
    np.testing.assert_allclose(head._all_anchors(sizes, dev).cpu().numpy(), anchors, rtol=0, atol=1e-4)
    cfg = configs.frcnn_train_cfg()['rpn']
    cls_r = [c.detach().cpu().double().requires_grad_(True) for c in cls]
    reg_r = [r.detach().cpu().double().requires_grad_(True) for r in reg]
    lc, lb, _ = ref.rpn_loss(cls_r, reg_r, anchors, gts, cfg['assigner'], cfg['sampler'], keys.numpy())
    (lc * 0.7 + lb * 1.3).backward()
    assert float(got['loss_rpn_cls'].detach()) == pytest.approx(float(lc), rel=1e-4)
    assert float(got['loss_rpn_bbox'].detach()) == pytest.approx(float(lb), rel=1e-4, abs=1e-9)
    for y, c, r in zip(head._raw_levels, cls_r, reg_r):
        g = y.grad.cpu().numpy()
        want = np.concatenate([c.grad.numpy(), r.grad.numpy()], 1)
        scale = np.abs(want).max() + 1e-12
        assert np.abs(g[:, :15] - want).max() <= 1e-3 * scale
        assert not np.any(g[:, 15:])                               # padding channel of the joint GEMM


@pytest.mark.parametrize('n_gt', [(6, 3), (0, 2), (0, 0), (60, 1)])
def test_roi_target_kernel_against_mmdet_checker(dev, ref, n_gt):
    m = _frcnn().to(dev)
    rng = np.random.default_rng(9)
    gts = [_boxes(rng, k, 1248, 384, 20, 200) for k in n_gt]
    gls = [rng.integers(0, 3, size=k) for k in n_gt]
    props = []
    for gt in gts:
        p = _boxes(rng, 1000, 1248, 384, 10, 150)
        if len(gt):
            src = rng.integers(0, len(gt), size=500)
            p[:500] = gt[src] + rng.normal(0, 6, size=(500, 4)).astype(np.float32)
        props.append((np.concatenate([p, rng.uniform(size=(1000, 1)).astype(np.float32)], 1), rng.uniform(size=1000) < 0.9))
    keys = rng.uniform(size=(2, 1000 + max(n_gt))).astype(np.float32)
    got = m.roi_head._targets_device([(torch.from_numpy(p).to(dev), torch.from_numpy(ok).to(dev)) for p, ok in props],
                                     [torch.from_numpy(g).to(dev) for g in gts],
                                     [torch.from_numpy(g).to(dev) for g in gls], torch.from_numpy(keys).to(dev))
    rois, labels, lw, tg, bw = [t.cpu().numpy() for t in got]
    live = lw > 0
    want = roi_target_rows(ref, props, gts, gls, keys)
    have = np.concatenate([rois, labels[:, None], lw[:, None], tg, bw], 1)[live]
    assert have.shape == want.shape
    a, b = sort_rows(have), sort_rows(want)
    assert np.array_equal(a[:, :7], b[:, :7]) and np.array_equal(a[:, 11:], b[:, 11:])     # rois, labels, weights
    np.testing.assert_allclose(a[:, 7:11], b[:, 7:11], rtol=1e-5, atol=1e-5)                # encoded deltas


def test_bbox_head_loss_kernel_against_mmdet_checker(dev, ref):
    m = _frcnn().to(dev)
    bh = m.roi_head.bbox_head
    g = torch.Generator().manual_seed(4)
    n = 1024
    labels = torch.randint(0, 4, (n,), generator=g)
    lw = (torch.rand(n, generator=g) < 0.8).float()
    tg = torch.randn(n, 4, generator=g)
    bw = (labels < 3).float()[:, None].expand(-1, 4).contiguous()
    cs0, bp0 = torch.randn(n, 4, generator=g) * 2, torch.randn(n, 12, generator=g)
    cs, bp = cs0.to(dev).requires_grad_(True), bp0.to(dev).requires_grad_(True)
    got = bh.loss(cs, bp, labels.to(dev), lw.to(dev), tg.to(dev), bw.to(dev))
    assert 'FusedBBoxHeadLoss' in type(got['loss_cls'].grad_fn.next_functions[0][0]).__name__
    (got['loss_cls'] * 1.3 + got['loss_bbox'] * 0.4).backward()
    keep = lw > 0
    cr, br = cs0[keep].double().requires_grad_(True), bp0[keep].double().requires_grad_(True)
    lc, lb, acc = ref.bbox_head_loss(cr, br, labels[keep], lw[keep], tg[keep], bw[keep], 3, alpha=0.5)
    (lc * 1.3 + lb * 0.4).backward()
    assert float(got['loss_cls'].detach()) == pytest.approx(float(lc), rel=1e-4)
    assert float(got['loss_bbox'].detach()) == pytest.approx(float(lb), rel=1e-4)
    assert float(got['acc']) == pytest.approx(float(acc), rel=1e-5)
    for gpu, want in ((cs.grad.cpu()[keep], cr.grad), (bp.grad.cpu()[keep], br.grad)):
        assert float((gpu.double() - want).abs().max()) <= 1e-3 * float(want.abs().max())
    assert float(cs.grad.cpu()[~keep].abs().max()) == 0.0


@pytest.mark.parametrize('phase', ['train', 'test'])
def test_rpn_proposal_kernels_against_mmdet_checker(dev, ref, phase):
    m = _head(dev)
    head = m.rpn_head
    feats = [torch.randn(2, 256, fh, fw, generator=torch.Generator().manual_seed(20 + i)).to(dev)
             .contiguous(memory_format=torch.channels_last) for i, (fh, fw) in enumerate(FULL)]
    metas = [dict(img_shape=(375, 1242, 3)), dict(img_shape=(384, 1248, 3))]
    cfg = m.train_cfg['rpn_proposal'] if phase == 'train' else m.test_cfg['rpn']
    with torch.no_grad():
        cls, reg = head(feats)
        got = head.get_bboxes(cls, reg, metas, cfg)
    anchors = ref.grid_anchors(FULL, STRIDES, [8], [0.5, 1.0, 2.0])
    for i, (props, ok) in enumerate(got):
        want = ref.rpn_get_bboxes_single([c[i].cpu().numpy() for c in cls], [r[i].cpu().numpy() for r in reg], anchors,
                                         metas[i]['img_shape'], cfg['nms_pre'], cfg['max_per_img'],
                                         cfg['nms']['iou_threshold'], cfg['min_bbox_size'])
        have = props[ok].cpu().numpy()
        assert have.shape == want.shape and len(want) > 100
        np.testing.assert_allclose(have[:, 4], want[:, 4], rtol=1e-5, atol=1e-6)           # same boxes, same order
        np.testing.assert_allclose(have[:, :4], want[:, :4], rtol=1e-5, atol=2e-3)


def test_detector_2d_step_against_mmdet_checker(dev, ref):
    """FasterRCNN.forward_train losses == the checker's chain on the same head outputs, proposals, keys
    (RoI features through the product's RoIAlign, which tests/test_ops_gpu.py pins to the oracle)."""
    from detmatch_amd import synth
    m = _frcnn().to(dev)
    with torch.no_grad():
        m.rpn_head.rpn_cls.weight.normal_(std=0.02)
        m.rpn_head.rpn_reg.weight.normal_(std=0.01)
    stu = synth.ssl_batch(2, 0, dev)['lab_stu']
    m.train()
    x = m.extract_feat(stu['img'])
    cls, reg = m.rpn_head(x)
    sizes = [tuple(c.shape[-2:]) for c in cls]
    n_anchor = sum(fh * fw * 3 for fh, fw in sizes)
    keys = torch.rand((2, n_anchor), generator=torch.Generator().manual_seed(5))
    losses = m.rpn_head.loss(cls, reg, stu['gt_bboxes'], stu['img_metas'], keys=keys.to(dev))
    anchors = ref.grid_anchors(sizes, STRIDES, [8], [0.5, 1.0, 2.0])
    cfg = configs.frcnn_train_cfg()
    gts = [g.cpu().numpy() for g in stu['gt_bboxes']]
    gls = [g.cpu().numpy() for g in stu['gt_labels']]
    lc, lb, _ = ref.rpn_loss([c.detach().cpu() for c in cls], [r.detach().cpu() for r in reg], np.concatenate(anchors),
                             gts, cfg['rpn']['assigner'], cfg['rpn']['sampler'], keys.numpy())
    assert float(losses['loss_rpn_cls'].detach()) == pytest.approx(float(lc), rel=1e-4)
    assert float(losses['loss_rpn_bbox'].detach()) == pytest.approx(float(lb), rel=1e-4)
    props = m.rpn_head.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg], stu['img_metas'],
                                  cfg['rpn_proposal'])
    k2 = torch.rand((2, props[0][0].shape[0] + max(len(g) for g in gts)), generator=torch.Generator().manual_seed(6))
    rois, labels, lw, tg, bw = m.roi_head._targets_device(props, stu['gt_bboxes'], stu['gt_labels'], k2.to(dev))
    want = roi_target_rows(ref, [(p.cpu().numpy(), ok.cpu().numpy()) for p, ok in props], gts, gls, k2.numpy())
    live = (lw > 0).cpu().numpy()
    have = np.concatenate([rois.cpu().numpy(), labels.cpu().numpy()[:, None], lw.cpu().numpy()[:, None],
                           tg.cpu().numpy(), bw.cpu().numpy()], 1)[live]
    assert have.shape == want.shape
    np.testing.assert_allclose(sort_rows(have), sort_rows(want), rtol=1e-5, atol=1e-5)
    cls_score, bbox_pred = m.roi_head.bbox_head(m.roi_head.extract(x, rois))
    got = m.roi_head.bbox_head.loss(cls_score, bbox_pred, labels, lw, tg, bw)
    keep = torch.from_numpy(live)
    l1, l2, acc = ref.bbox_head_loss(cls_score.detach().cpu()[keep], bbox_pred.detach().cpu()[keep], labels.cpu()[keep],
                                     lw.cpu()[keep], tg.cpu()[keep], bw.cpu()[keep], 3, alpha=0.5)
    assert float(got['loss_cls'].detach()) == pytest.approx(float(l1), rel=1e-4)
    assert float(got['loss_bbox'].detach()) == pytest.approx(float(l2), rel=1e-4, abs=1e-9)
    assert float(got['acc']) == pytest.approx(float(acc), rel=1e-5)
