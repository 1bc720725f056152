"""Fused assignment / sampling / loss kernels of the 2-D detector (csrc/det2d_targets.hip) against the
dense tensor formulation of the same rules (mm2d/faster_rcnn.py, itself checked against a brute-force
MaxIoU definition in test_frcnn_gpu.py) with the same random keys: sampled sets, targets, loss values
and the gradients that reach the head outputs."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


def _frcnn(dev):
    from detmatch_amd.mm2d.faster_rcnn import FasterRCNN
    cfg = configs.frcnn_kitti_model()
    cfg.pop('type')
    torch.manual_seed(0)
    return FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg).to(dev)


def _boxes(rng, n, w, h, lo=20, hi=200):
    c = rng.uniform([0, 0], [w, h], size=(n, 2))
    s = rng.uniform(lo, hi, size=(n, 2))
    b = np.concatenate([np.clip(c - s / 2, 0, None), np.minimum(c + s / 2, [w, h])], 1)
    return torch.from_numpy(b.astype(np.float32))


@pytest.mark.parametrize('n_gt', [(6, 3), (1, 0), (0, 0), (40, 17)])
def test_rpn_loss_kernel_matches_tensor_formulation(dev, n_gt):
    m = _frcnn(dev)
    head = m.rpn_head
    with torch.no_grad():                                   # informative logits / deltas
        head.rpn_cls.weight.normal_(std=0.05)
        head.rpn_reg.weight.normal_(std=0.02)
    rng = np.random.default_rng(5)
    sizes = [(96, 312), (48, 156), (24, 78), (12, 39), (6, 20)]
    g = torch.Generator().manual_seed(1)
    gts = [_boxes(rng, k, 1248, 384).to(dev) for k in n_gt]
    res = []
    for fused in (True, False):
        feats = [torch.randn(2, 256, h, w, generator=torch.Generator().manual_seed(10 + i))
                 .to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                 for i, (h, w) in enumerate(sizes)]
        cls, reg = head(feats)
        n_anchor = sum(h * w * 3 for h, w in sizes)
        keys = torch.rand((2, n_anchor), generator=torch.Generator().manual_seed(77)).to(dev)
        losses = head.loss(cls, reg, gts, None, fused=fused, keys=keys)
        assert ('FusedRpnLoss' in type(losses['loss_rpn_cls'].grad_fn.next_functions[0][0]).__name__) == fused
        (losses['loss_rpn_cls'] * 0.7 + losses['loss_rpn_bbox'] * 1.9).backward()
        res.append((losses, [f.grad.clone() for f in feats], head.rpn_conv.weight.grad.clone()))
        head.zero_grad()
    a, b = res
    close(a[0]['loss_rpn_cls'], b[0]['loss_rpn_cls'], rtol=2e-5)
    close(a[0]['loss_rpn_bbox'], b[0]['loss_rpn_bbox'], rtol=2e-5)
    if max(n_gt) == 0:
        assert float(a[0]['loss_rpn_bbox']) == 0.0
    for ga, gb in zip(a[1], b[1]):
        close(ga, gb, rtol=1e-3, atol=1e-7)
    close(a[2], b[2], rtol=1e-3, atol=1e-6)


def _proposals(rng, dev, gts, n=1000):
    out = []
    for gt in gts:
        p = _boxes(rng, n, 1248, 384, 10, 150)
        k = gt.shape[0]
        if k:                                                # half of them jittered copies of GT boxes
            src = torch.from_numpy(rng.integers(0, k, size=n // 2))
            p[: n // 2] = gt.cpu()[src] + torch.from_numpy(rng.normal(0, 6, size=(n // 2, 4)).astype(np.float32))
        score = torch.from_numpy(rng.uniform(size=(n, 1)).astype(np.float32))
        ok = torch.from_numpy(rng.uniform(size=n) < 0.9)
        out.append((torch.cat([p, score], 1).to(dev), ok.to(dev)))
    return out


@pytest.mark.parametrize('n_gt', [(6, 3), (0, 2), (0, 0), (60, 1)])
def test_roi2d_target_kernel_matches_tensor_formulation(dev, n_gt):
    m = _frcnn(dev)
    rh = m.roi_head
    rng = np.random.default_rng(9)
    gts = [_boxes(rng, k, 1248, 384).to(dev) for k in n_gt]
    gls = [torch.from_numpy(rng.integers(0, 3, size=k)).to(dev) for k in n_gt]
    props = _proposals(rng, dev, gts)
    keys = torch.rand((2, 1000 + max(n_gt)), generator=torch.Generator().manual_seed(3)).to(dev)
    a = rh._targets_device(props, gts, gls, keys)
    b = rh._targets_tensor(props, gts, gls, keys)
    names = ('rois', 'labels', 'label_weights', 'bbox_targets', 'bbox_weights')
    for name, x, y in zip(names, a, b):
        assert x.shape == y.shape and x.dtype == y.dtype, name
        if name == 'bbox_targets':
            close(x, y, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(x, y), name
    lw = a[2].view(2, 512)
    for i, k in enumerate(n_gt):
        n_pos = int((a[1].view(2, 512)[i] < 3).sum())
        assert n_pos <= 128 and (k == 0) == (n_pos == 0)
        assert int(lw[i].sum()) == 512                       # 1000 + k boxes: the sampler always fills up


def test_bbox_head_loss_kernel_matches_tensor_formulation(dev):
    m = _frcnn(dev)
    bh = m.roi_head.bbox_head
    g = torch.Generator().manual_seed(4)
    n = 1024
    labels = torch.randint(0, 4, (n,), generator=g).to(dev)
    lw = (torch.rand(n, generator=g) < 0.8).float().to(dev)
    tg = torch.randn(n, 4, generator=g).to(dev)
    bw = (labels < 3).float()[:, None].expand(-1, 4).contiguous()
    res = []
    for fused in (True, False):
        cs = (torch.randn(n, 4, generator=torch.Generator().manual_seed(5)) * 2).to(dev).requires_grad_(True)
        bp = torch.randn(n, 12, generator=torch.Generator().manual_seed(6)).to(dev).requires_grad_(True)
        losses = bh.loss(cs, bp, labels, lw, tg, bw, fused=fused)
        (losses['loss_cls'] * 1.3 + losses['loss_bbox'] * 0.4).backward()
        res.append((losses, cs.grad.clone(), bp.grad.clone()))
    a, b = res
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        close(a[0][k], b[0][k], rtol=2e-5)
    close(a[1], b[1], rtol=1e-4, atol=1e-9)
    close(a[2], b[2], rtol=1e-4, atol=1e-9)


def test_faster_rcnn_step_fused_equals_tensor_path(dev):
    """Whole forward_train: the same seed gives the same random keys on both paths."""
    from detmatch_amd import synth
    m = _frcnn(dev)
    data = synth.ssl_batch(2, 0, dev)
    stu = data['lab_stu']
    m.train()
    out = []
    for fused in (True, False):
        torch.manual_seed(11)
        x = m.extract_feat(stu['img'])
        cls, reg = m.rpn_head(x)
        losses = m.rpn_head.loss(cls, reg, stu['gt_bboxes'], stu['img_metas'], fused=fused)
        props = m.rpn_head.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg], stu['img_metas'],
                                      m.train_cfg['rpn_proposal'])
        losses.update(m.roi_head.forward_train(x, stu['img_metas'], props, stu['gt_bboxes'], stu['gt_labels'],
                                               fused=fused))
        m.zero_grad()
        sum(v for k, v in losses.items() if 'loss' in k).backward()
        out.append((losses, m.neck.lateral_convs[0].conv.weight.grad.clone(),
                    m.roi_head.bbox_head.fc_reg.weight.grad.clone()))
    for k in out[0][0]:
        close(out[0][0][k], out[1][0][k], rtol=1e-4, atol=1e-6)
    close(out[0][1], out[1][1], rtol=2e-3, atol=1e-6)
    close(out[0][2], out[1][2], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize('phase', ['train', 'test'])
def test_rpn_proposal_kernel_matches_tensor_formulation(dev, phase):
    """Per-level top-k + decode + batched-NMS inputs in two launches == the per-image tensor chain:
    the same proposals in the same order."""
    m = _frcnn(dev)
    head = m.rpn_head
    with torch.no_grad():
        head.rpn_cls.weight.normal_(std=0.05)
        head.rpn_reg.weight.normal_(std=0.03)
    sizes = [(96, 312), (48, 156), (24, 78), (12, 39), (6, 20)]
    feats = [torch.randn(2, 256, h, w, generator=torch.Generator().manual_seed(20 + i))
             .to(dev).contiguous(memory_format=torch.channels_last) for i, (h, w) in enumerate(sizes)]
    metas = [dict(img_shape=(375, 1242, 3)), dict(img_shape=(384, 1248, 3))]
    cfg = m.train_cfg['rpn_proposal'] if phase == 'train' else m.test_cfg['rpn']
    with torch.no_grad():
        cls, reg = head(feats)
        a = head.get_bboxes(cls, reg, metas, cfg)
        b = head.get_bboxes(cls, reg, metas, cfg, fused=False)
        pre = head._pre_nms_device(head._raw_levels, [c.shape[-2:] for c in cls], metas, cfg)
    t = sum(min(cfg['nms_pre'], h * w * 3) for h, w in sizes)
    assert pre[0].shape == (2, t, 4) and bool(pre[2].any())
    for (pa, oka), (pb, okb) in zip(a, b):
        assert pa.shape == (cfg['max_per_img'], 5) and int(oka.sum()) > 100
        assert torch.equal(oka, okb)
        assert torch.equal(pa, pb)
