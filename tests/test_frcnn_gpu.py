"""GPU parity/behaviour tests of the 2D branch (SURVEY §8 G — third party, parity unpinned:
the HIP ops are checked against the oracle restatement; the detector against invariants)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rois(rng, n, w, h, batch):
    c = rng.uniform([0, 0], [w, h], size=(n, 2))
    s = rng.uniform(4, 0.6 * w, size=(n, 2)) * rng.uniform(0.05, 1, size=(n, 1))
    b = np.concatenate([c - s / 2, c + s / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, w)
    b[:, 1::2] = b[:, 1::2].clip(0, h)
    return np.concatenate([rng.integers(0, batch, size=(n, 1)).astype(np.float64), b], 1).astype(np.float32)


@pytest.mark.parametrize('aligned,sampling_ratio', [(True, 0), (False, 0), (True, 2)])
def test_roi_align_single_level_matches_oracle(dev, orc, aligned, sampling_ratio):
    from detmatch_amd.roi_align import roi_align
    rng = np.random.default_rng(3)
    feat = rng.normal(size=(2, 16, 24, 40)).astype(np.float32)
    rois = _rois(rng, 60, 160, 96, 2)
    rois[0, 1:] = [-30, -20, 400, 300]        # far outside: taps flagged invalid / clamped
    rois[1, 1:] = [50, 40, 50, 40]            # empty box
    f = torch.from_numpy(feat).to(dev).requires_grad_()
    out = roi_align(f, torch.from_numpy(rois).to(dev), 7, 0.25, sampling_ratio, aligned)
    want = orc.roi_align(feat, rois, 0.25, 7, sampling_ratio, aligned)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    g = rng.normal(size=want.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    gw = orc.roi_align_grad(g, feat.shape, rois, 0.25, sampling_ratio, aligned)
    np.testing.assert_allclose(f.grad.cpu().numpy(), gw, rtol=1e-4, atol=1e-4)


def test_roi_align_fpn_matches_per_level_oracle(dev, orc):
    """One launch over 4 levels == the per-level masked evaluation of SingleRoIExtractor."""
    from detmatch_amd.roi_align import map_roi_levels, roi_align_fpn
    rng = np.random.default_rng(5)
    strides = [4, 8, 16, 32]
    feats = [rng.normal(size=(2, 8, 96 // (s // 4), 320 // (s // 4))).astype(np.float32) for s in strides]
    rois = _rois(rng, 300, 1280, 384, 2)
    tf = [torch.from_numpy(f).to(dev).requires_grad_() for f in feats]
    tr = torch.from_numpy(rois).to(dev)
    out = roi_align_fpn(tf, tr, strides)
    lv = map_roi_levels(tr, 4).cpu().numpy()
    assert set(np.unique(lv)) == {0, 1, 2, 3}
    want = np.zeros((300, 8, 7, 7), np.float32)
    for l, s in enumerate(strides):
        m = lv == l
        want[m] = orc.roi_align(feats[l], rois[m], 1.0 / s, 7, 0, True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    g = rng.normal(size=want.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    for l, s in enumerate(strides):
        m = lv == l
        gw = orc.roi_align_grad(g[m], feats[l].shape, rois[m], 1.0 / s, 0, True)
        np.testing.assert_allclose(tf[l].grad.cpu().numpy(), gw, rtol=1e-4, atol=1e-4)


def test_roi_align_backward_wide_roi_and_nchw_entry(dev, orc):
    """RoIs wider than the LDS tap tables (> 160 px span) take the in-kernel per-sample path; the
    NCHW entry point dm_roi_align_backward stays available to C callers."""
    import ctypes
    from detmatch_amd import _lib
    from detmatch_amd.roi_align import roi_align
    rng = np.random.default_rng(9)
    feat = rng.normal(size=(1, 8, 170, 180)).astype(np.float32)
    rois = np.array([[0, 2, 3, 178, 168], [0, 20, 30, 60, 90], [0, -5, -5, 400, 400]], np.float32)
    f = torch.from_numpy(feat).to(dev).requires_grad_()
    tr = torch.from_numpy(rois).to(dev)
    out = roi_align(f, tr, 7, 1.0, 0, True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), orc.roi_align(feat, rois, 1.0, 7, 0, True),
                               rtol=1e-5, atol=1e-5)
    g = rng.normal(size=(3, 8, 7, 7)).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    want = orc.roi_align_grad(g, feat.shape, rois, 1.0, 0, True)
    np.testing.assert_allclose(f.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    # NCHW C entry
    gn = torch.zeros(feat.shape, device=dev)
    tg = torch.from_numpy(g).to(dev)
    ptrs = (ctypes.c_void_p * 1)(gn.data_ptr())
    hs, ws, sc = (ctypes.c_int32 * 1)(170), (ctypes.c_int32 * 1)(180), (ctypes.c_float * 1)(1.0)
    _lib.check(_lib.lib().dm_roi_align_backward(ptrs, hs, ws, sc, 1, 8, _lib.ptr(tr), None, 3, 7, 7, 0, 1,
                                                8, _lib.ptr(tg), _lib.stream()), 'dm_roi_align_backward')
    np.testing.assert_allclose(gn.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    on = torch.empty((3, 8, 7, 7), device=dev)
    fp = (ctypes.c_void_p * 1)(f.data_ptr())
    _lib.check(_lib.lib().dm_roi_align_forward(fp, hs, ws, sc, 1, 8, _lib.ptr(tr), None, 3, 7, 7, 0, 1, 8,
                                               _lib.ptr(on), _lib.stream()), 'dm_roi_align_forward')
    np.testing.assert_allclose(on.cpu().numpy(), orc.roi_align(feat, rois, 1.0, 7, 0, True),
                               rtol=1e-5, atol=1e-5)


def test_nms_2d_matches_numpy(dev):
    from detmatch_amd.mm2d.faster_rcnn import nms_fixed
    rng = np.random.default_rng(11)
    c = rng.uniform(0, 300, size=(700, 2))
    s = rng.uniform(10, 80, size=(700, 2))
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    scores = rng.uniform(size=700).astype(np.float32)
    idx, ok = nms_fixed(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.5, 200)
    order = np.argsort(-scores, kind='stable')
    keep, b = [], boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(700, bool)
    for i in range(700):
        if sup[i]:
            continue
        keep.append(order[i])
        lt = np.maximum(b[i, :2], b[:, :2])
        rb = np.minimum(b[i, 2:], b[:, 2:])
        wh = np.clip(rb - lt, 0, None)
        inter = wh[:, 0] * wh[:, 1]
        sup |= inter / (area[i] + area - inter) > 0.5
    keep = np.array(keep[:200])
    n = int(ok.sum())
    assert n == len(keep) and np.array_equal(idx[:n].cpu().numpy(), keep)


def _frcnn(dev):
    from detmatch_amd import configs
    from detmatch_amd.mm2d import FasterRCNN
    cfg = configs.frcnn_kitti_model()
    cfg.pop('type')
    torch.manual_seed(0)
    return FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg).to(dev)


def test_faster_rcnn_train_and_test_paths(dev):
    from detmatch_amd import synth
    m = _frcnn(dev)
    data = synth.ssl_batch(2, 0, dev)
    stu = data['lab_stu']
    m.train()
    losses = m.forward_train(stu['img'], stu['img_metas'], stu['gt_bboxes'], stu['gt_labels'])
    assert set(losses) == {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'acc', 'loss_bbox'}
    for k, v in losses.items():
        assert v.dim() == 0 and torch.isfinite(v), k
    # untrained RPN: BCE of ~0 logits = ln 2
    assert 0.3 < float(losses['loss_rpn_cls'].detach()) < 2.0
    total = sum(v for k, v in losses.items() if 'loss' in k)
    total.backward()
    assert m.backbone.conv1.weight.grad is None and m.backbone.layer1[0].conv1.weight.grad is None
    for p in (m.backbone.layer2[0].downsample[0].weight, m.neck.lateral_convs[0].conv.weight,
              m.rpn_head.rpn_conv.weight, m.roi_head.bbox_head.fc_reg.weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    # zero GT (empty pseudo-label set) must train too
    z = m.forward_train(stu['img'], stu['img_metas'], [stu['gt_bboxes'][0][:0]] * 2,
                        [stu['gt_labels'][0][:0]] * 2)
    assert float(z['loss_bbox']) == 0.0 and float(z['loss_rpn_bbox']) == 0.0 and torch.isfinite(z['loss_cls'])
    m.eval()
    with torch.no_grad():
        res = m.simple_test_pre_nms(stu['img'], stu['img_metas'])
    assert len(res) == 2
    boxes, scores = res[0]
    assert boxes.shape == (1000, 12) and scores.shape == (1000, 4)
    assert float(boxes.min()) >= 0 and float(boxes[:, 0::4].max()) <= 1280 and float(boxes[:, 1::4].max()) <= 384
    assert float(scores.min()) >= 0 and float(scores.max()) <= 1


def test_rpn_targets_against_brute_force(dev):
    """Dense MaxIoU assignment == the reference's sequential definition on a small case."""
    from detmatch_amd.mm2d.faster_rcnn import max_iou_assign
    from detmatch_amd.mm3d.losses import bbox_overlaps
    rng = np.random.default_rng(2)
    c = rng.uniform(0, 200, size=(400, 2)); s = rng.uniform(10, 90, size=(400, 2))
    boxes = torch.from_numpy(np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)).to(dev)
    gt = boxes[rng.permutation(400)[:7]] + 3
    got = max_iou_assign(boxes, gt, 0.7, 0.3, 0.3, True).cpu().numpy()
    ov = bbox_overlaps(gt, boxes).cpu().numpy()
    want = np.full(400, -1)
    mx, am = ov.max(0), ov.argmax(0)
    want[(mx >= 0) & (mx < 0.3)] = 0
    want[mx >= 0.7] = am[mx >= 0.7] + 1
    for i in range(7):
        if ov[i].max() >= 0.3:
            want[ov[i] == ov[i].max()] = i + 1
    assert np.array_equal(got, want)
    assert (max_iou_assign(boxes, gt[:0], 0.7, 0.3, 0.3, True) == 0).all()


def test_backbone_and_fpn_epilogue_sums_are_the_separate_kernels(dev):
    """ResNet-50 + FPN with the shortcut / top-down sums in the convolutions' epilogues (fused switch on)
    == the same network with separate add and ReLU kernels: feature maps bit for bit, weight gradients
    equal up to the summation order of the gradient that reaches a tensor along two paths."""
    from detmatch_amd import configs, fused
    from detmatch_amd.mm2d.faster_rcnn import FasterRCNN
    cfg = configs.frcnn_kitti_model()
    cfg.pop('type')
    torch.manual_seed(0)
    m = FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg).to(dev).train()
    with torch.no_grad():      # zero_init_residual would hide the last BN of every block
        for mod in m.backbone.modules():
            if hasattr(mod, 'bn3'):
                mod.bn3.weight.fill_(0.7)
    img = torch.randn(2, 3, 96, 160, generator=torch.Generator().manual_seed(1)).to(dev)
    out = []
    prev = fused.ENABLED
    try:
        for on in (True, False):
            fused.ENABLED = on
            m.zero_grad()
            feats = m.extract_feat(img)
            gs = [torch.randn(f.shape, generator=torch.Generator().manual_seed(7 + i)).to(dev)
                  for i, f in enumerate(feats)]
            torch.autograd.backward(list(feats), gs)
            out.append(([f.detach().clone() for f in feats],
                        m.backbone.layer3[2].conv3.weight.grad.clone(),
                        m.backbone.layer2[0].conv1.weight.grad.clone(),
                        m.neck.lateral_convs[1].conv.weight.grad.clone()))
    finally:
        fused.ENABLED = prev
    (fa, *ga), (fb, *gb) = out
    for a, b in zip(fa, fb):
        assert torch.equal(a, b)
    for a, b in zip(ga, gb):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale
