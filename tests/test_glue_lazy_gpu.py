"""The pseudo-label glue without compaction read-backs (ssl_modules.Masked, bbox_utils.filter_by_nms_2d_masked, the
batched FusionHungarianMatching): same box lists, same pseudo labels, same losses as the module-by-module,
sample-by-sample compaction of the reference (processors_fusion.py:29-46 / :97-222, bbox_utils.py:282-347)."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _dets(dev, seed, n, c, bg):
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(n, 2, generator=g) * torch.tensor([1242.0, 375.0])
    wh = torch.rand(n, 2, generator=g) * torch.tensor([260.0, 140.0]) + 4
    boxes = torch.cat([(ctr - wh / 2).clamp(min=0), torch.minimum(ctr + wh / 2, torch.tensor([1242.0, 375.0]))], 1)
    s = torch.rand(n, c + int(bg), generator=g) ** 3
    if bg:
        s = s / s.sum(1, keepdim=True)
    return boxes.to(dev), s.to(dev)


@pytest.mark.parametrize('n,c,bg,thr,max_num,masked', [(1000, 3, True, 0.05, 100, False), (300, 3, False, 0.1, 100, True),
                                                        (40, 3, True, 0.2, 100, False), (64, 3, False, 0.999, 100, True),
                                                        (500, 3, True, 0.01, 17, True), (1, 3, False, 0.0, 100, False)])
def test_masked_nms_equals_compacting_nms(dev, n, c, bg, thr, max_num, masked):
    from detmatch_amd.mm3d.bbox_utils import filter_by_nms_2d, filter_by_nms_2d_masked, take
    cfg = dict(score_thr=thr, nms_pre=-1, max_num=max_num, iou_thr=0.5)
    boxes, scores = _dets(dev, n * 7 + c, n, c, bg)
    keep_in = None
    if masked:
        keep_in = torch.rand(n, generator=torch.Generator().manual_seed(n)).to(dev) > 0.3
    if keep_in is not None:
        ki = keep_in.nonzero().squeeze(1)
        want = filter_by_nms_2d([(take(boxes, ki), take(scores, ki))], cfg, not bg)[0]
    else:
        want = filter_by_nms_2d([(boxes, scores)], cfg, not bg)[0]
    if n == 300:                      # one box per class (the teacher's bbox head, reg_class_agnostic=False)
        boxes = torch.cat([boxes, boxes + 3.0, boxes * 0.9], dim=1)
        want = filter_by_nms_2d([(take(boxes, ki), take(scores, ki))], cfg, not bg)[0]
    (got_b, got_s), ok = filter_by_nms_2d_masked([((boxes, scores), keep_in)], cfg, not bg)[0]
    assert got_b.shape[0] == min(max_num, n * c) and ok.dtype == torch.bool
    k = int(ok.sum())
    assert bool(ok[:k].all()) and k == want[0].shape[0]           # valid rows first, as many as the reference keeps
    assert torch.equal(got_b[:k], want[0]) and torch.equal(got_s[:k], want[1])


def _chain(dev):
    """The first glue block of the DetMatch recipe on synthetic teacher outputs (modules 01-12 of
    configs.detmatch_ssl_cfg, without the two detector passes)."""
    from detmatch_amd.mm3d.registry import SSL_MODULES, build_from_cfg
    cfg = configs.detmatch_ssl_cfg(with_vis=False)['unlabeled']
    skip = ('Opd_SimpleTest_3D', 'SimpleTest_2D', 'Opd_HardPseudoLabel_3D', 'HardPseudoLabel_2D')
    first = []
    for m in cfg:
        if m['type'] == 'Opd_HardPseudoLabel_3D':
            break
        if m['type'] not in skip:
            first.append(build_from_cfg(dict(m), SSL_MODULES))
    return first


def _batch(dev, seed):
    from test_ssl_match_gpu import _scene
    out = dict(tea=dict(img_metas=[], **{'3d_bboxes_nms': [], '2d_bboxes': []}), stu=dict(img_metas=[]))
    for b in range(2):
        boxes3, s3, b2, s2, meta = _scene(dev, seed + b, 60 + 20 * b, 1)
        pb, ps = _dets(dev, seed * 3 + b, 800, 3, True)
        # 2D candidates near the projections of some 3D boxes, so that the matching has something to match
        from detmatch_amd.mm3d.bbox_utils import bbox_3d_to_bbox_2d
        proj, valid = bbox_3d_to_bbox_2d(boxes3, meta['lidar2img'], meta['ori_shape'])
        pb[:proj.shape[0]] = proj + (torch.randn(proj.shape, generator=torch.Generator().manual_seed(seed + 50 + b)) * 3).to(dev)
        ps[:proj.shape[0], :3] = s3
        ps[:proj.shape[0], 3] = 0.02
        tea_meta = dict(meta, scale_factor=np.array([1.03, 1.03, 1.03, 1.03], np.float32), flip=bool(b),
                        img_shape=(386, 1279, 3), transformation_3d_flow=['HF'] if b else [], pcd_horizontal_flip=bool(b))
        stu_meta = dict(tea_meta, transformation_3d_flow=tea_meta['transformation_3d_flow'] + ['R', 'S', 'T'],
                        pcd_rotation=np.array([[0.98, 0.199, 0], [-0.199, 0.98, 0], [0, 0, 1]], np.float32),
                        pcd_scale_factor=1.02, pcd_trans=np.zeros(3, np.float32))
        out['tea']['img_metas'].append(tea_meta)
        out['stu']['img_metas'].append(stu_meta)
        out['tea']['3d_bboxes_nms'].append((boxes3, s3))
        out['tea']['2d_bboxes'].append((pb, ps))
    return out


def test_first_glue_block_lazy_equals_compacting(dev, monkeypatch):
    from detmatch_amd.mm3d import ssl_modules as M
    res = []
    for lazy in (True, False):
        if not lazy:
            monkeypatch.setattr(M, '_lazy', lambda t: False)
        bd = _batch(dev, 11)
        for m in _chain(dev):
            bd = m.forward(None, bd)
        res.append(bd)
    a, b = res
    keys = [k for k in a['tea'] if k.endswith('_hung') or k.endswith('_dtch')]
    assert len(keys) >= 4
    for k in keys:
        for ea, eb in zip(M.plain(a['tea'][k]), M.plain(b['tea'][k])):
            for ta, tb in zip(ea, eb):
                ta = ta.tensor if hasattr(ta, 'tensor') else ta
                tb = tb.tensor if hasattr(tb, 'tensor') else tb
                assert ta.shape == tb.shape and torch.equal(ta, tb), k
    assert sum(len(e[0]) for e in a['tea']['3d_bboxes_nms_no_aug_hung']) > 0          # something was matched
    # the lazy run kept its filters as masks up to the matching
    assert isinstance(a['tea']['2d_bboxes_nms'][0], M.Masked) and isinstance(a['tea']['3d_bboxes_nms_no_aug_sc_filt'][0], M.Masked)
    assert not isinstance(b['tea']['2d_bboxes_nms'][0], M.Masked)


def test_iteration_with_lazy_glue_equals_compacting_glue(dev, monkeypatch):
    """The real DetMatch iteration: same losses, same accumulated gradient with the masks as with the compactions
    (dense pseudo ground truth for the 3D student included)."""
    from detmatch_amd.mm3d import ssl_modules as M
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    for lazy in (True, False):
        if not lazy:
            monkeypatch.setattr(M, '_lazy', lambda t: False)
        wl = DetMatchTrainWorkload(2, dev)
        torch.manual_seed(321)
        wl.step()
        torch.cuda.synchronize()
        out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
        del wl
    (ga, la), (gb, lb) = out
    assert set(la) == set(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    assert float(la.get('ssl.unlab.metrics.num_tea_hung', 1.0)) >= 0
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 1e-3, float(rel)
