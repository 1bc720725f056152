"""PV-RCNN module graph on the native ops: training step, eval forward, empty-target edge
cases (zero pseudo-labels is a normal case and must not crash or NaN)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev):
    from detmatch_amd import configs
    from detmatch_amd.mm3d.openpcdet import OpenPCDetDetector
    torch.manual_seed(0)
    cfg = configs.pvrcnn_kitti_model()
    cfg.pop('type')
    return OpenPCDetDetector(**cfg).to(dev)


def _batch(dev, seeds=(0, 1)):
    from detmatch_amd import synth
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    frames = [synth.lidar_frame(s) for s in seeds]
    pts = [torch.from_numpy(f['points']).to(dev) for f in frames]
    boxes, labels = [], []
    for f in frames:
        b, l = synth.frame_to_mm3d_gt(f)
        boxes.append(LiDARInstance3DBoxes(torch.from_numpy(b).to(dev)))
        labels.append(torch.from_numpy(l).to(dev))
    metas = [dict(sample_idx=i) for i in range(len(frames))]
    return pts, metas, boxes, labels


def test_box_convention_roundtrip(dev):
    from detmatch_amd import synth
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    from detmatch_amd.mm3d.openpcdet import mm3d_to_pcdet_boxes, pcdet_to_mm3d_boxes
    f = synth.lidar_frame(0)
    b, _ = synth.frame_to_mm3d_gt(f)
    pc = mm3d_to_pcdet_boxes(LiDARInstance3DBoxes(torch.from_numpy(b)))
    want = f['gt_boxes'].copy()
    got = pc.numpy()
    np.testing.assert_allclose(got[:, :6], want[:, :6], atol=1e-5)
    d = (got[:, 6] - want[:, 6] + np.pi) % (2 * np.pi) - np.pi
    np.testing.assert_allclose(d, 0, atol=1e-5)
    back = pcdet_to_mm3d_boxes(pc).tensor.numpy()
    np.testing.assert_allclose(back[:, :6], b[:, :6], atol=1e-5)


def test_train_step_and_eval(dev):
    det = _build(dev)
    pts, metas, boxes, labels = _batch(dev)
    det.train()
    out = det.forward_train(pts, metas, boxes, labels)
    loss = out['loss']
    assert torch.isfinite(loss)
    loss.backward()
    n_grad = 0
    for name, p in det.named_parameters():
        assert p.grad is not None, name
        assert torch.isfinite(p.grad).all(), name
        n_grad += int((p.grad != 0).any())
    assert n_grad > 0.9 * len(list(det.parameters()))
    fr = det.model.roi_head.forward_ret_dict
    assert fr['rois'].shape == (2, 128, 7) and fr['rcnn_cls'].shape[0] == 256
    # second step with an optimizer: loss moves, still finite
    opt = torch.optim.AdamW(det.parameters(), lr=1e-3)
    opt.step()
    opt.zero_grad()
    loss2 = det.forward_train(pts, metas, boxes, labels)['loss']
    assert torch.isfinite(loss2)
    det.eval()
    with torch.no_grad():
        res = det.simple_test(pts, metas)
    assert len(res) == 2
    for r in res:
        n = len(r['scores_3d'])
        assert r['boxes_3d'].tensor.shape == (n, 7) and r['labels_3d'].shape == (n,)
        assert n <= 100


def test_empty_targets(dev):
    """No GT at all / GT of out-of-range classes only: every anchor background, single zero GT
    in the proposal layer, nothing NaN."""
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    det = _build(dev)
    pts, metas, boxes, labels = _batch(dev)
    empty_boxes = [LiDARInstance3DBoxes(torch.zeros((0, 7), device=dev)) for _ in pts]
    empty_labels = [torch.zeros((0,), dtype=torch.long, device=dev) for _ in pts]
    det.train()
    loss = det.forward_train(pts, metas, empty_boxes, empty_labels)['loss']
    assert torch.isfinite(loss)
    loss.backward()
    for name, p in det.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
    bad_labels = [torch.full_like(l, 7) for l in labels]
    loss = det.forward_train(pts, metas, boxes, bad_labels)['loss']
    assert torch.isfinite(loss)
    one = [boxes[0], empty_boxes[1]]
    onel = [labels[0], empty_labels[1]]
    assert torch.isfinite(det.forward_train(pts, metas, one, onel)['loss'])


def test_pvrcnn_supervised_loss_decreases(dev):
    """End-to-end sanity of every gradient path (dense conv dgrad / wgrad, sparse conv, BN rows, RoI
    head, fused AdamW): 25 supervised PV-RCNN steps on one fixed synthetic batch reduce the loss."""
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import PVRCNNTrainWorkload
    wl = PVRCNNTrainWorkload([synth.lidar_frame(s) for s in (0, 1)], dev)
    losses = [float(wl.step()) for _ in range(25)]
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:3]), losses
