"""Generates tests/golden/hungarian.npz by running the REFERENCE's ModHungarianAssigner.assign
and DoubleSidedFocalLossCost (loaded by file path from /root/reference; build container only)
with the real scipy.optimize.linear_sum_assignment.

mmdet (2.14.0, un-vendored) is absent, so the three mmdet cost classes the reference imports
(FocalLossCost / BBoxL1Cost / IoUCost), bbox_cxcywh_to_xyxy and AssignResult are supplied from
this repo's restatement (detmatch_amd/mm3d/losses.py, SURVEY §8a-G formulas).  What the
fixture therefore pins is the reference's OWN logic: the cost sum, LAP, 1-based gt_inds,
max_overlaps = matched cost / Inf, iou_cost, the empty-input early return and the double-sided
focal cost; the three inner cost formulas stay "parity unpinned".

The fixture also stores each case's summed fp32 cost matrix and scipy's assignment on it,
which pins dm_lap_host independently of any cost formula.

    python tests/golden/gen_hungarian_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(dotted, rel):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    from detmatch_amd.mm3d import losses as L
    from detmatch_amd.mm3d.registry import Registry, build_from_cfg
    match_cost = Registry('match_cost_ref')
    for c in (L.FocalLossCost, L.BBoxL1Cost, L.IoUCost):
        match_cost.register_module(c)
    assigners = Registry('assigner_ref')
    for n in ('mmdet', 'mmdet.core', 'mmdet.core.bbox', 'mmdet.core.bbox.assigners'):
        _stub(n)
    _stub('mmdet.core.bbox.builder', BBOX_ASSIGNERS=assigners)
    _stub('mmdet.core.bbox.match_costs', build_match_cost=lambda cfg: build_from_cfg(cfg, match_cost),
          FocalLossCost=L.FocalLossCost)
    _stub('mmdet.core.bbox.match_costs.builder', MATCH_COST=match_cost)
    _stub('mmdet.core.bbox.transforms', bbox_cxcywh_to_xyxy=L.bbox_cxcywh_to_xyxy,
          bbox_xyxy_to_cxcywh=L.bbox_xyxy_to_cxcywh)
    _stub('mmdet.core.bbox.iou_calculators', bbox_overlaps=L.bbox_overlaps)
    _stub('mmdet.core.bbox.assigners.assign_result', AssignResult=L.AssignResult)
    _stub('mmdet.core.bbox.assigners.base_assigner', BaseAssigner=object)
    _load('ref_modified_match_cost', 'mmdet3d/core/bbox/match_costs/modified_match_cost.py')
    mod = _load('ref_modified_hungarian_assigner',
                'mmdet3d/core/bbox/assigners/modified_hungarian_assigner.py')
    return mod.ModHungarianAssigner


# configs/detmatch/001/detmatch/split_0.py:260-268
ASSIGNER_CFG = dict(cls_cost=dict(type='DoubleSidedFocalLossCost', weight=2.0),
                    reg_cost=dict(type='BBoxL1Cost', weight=5.0),
                    iou_cost=dict(type='IoUCost', iou_mode='giou', weight=2.0))


def random_case(rng, n_pred, n_gt, w=1242, h=375):
    def boxes(n):
        c = rng.uniform([0, 0], [w, h], size=(n, 2))
        s = rng.uniform(8, 200, size=(n, 2))
        b = np.concatenate([c - s / 2, c + s / 2], axis=1)
        b[:, 0::2] = b[:, 0::2].clip(0, w)
        b[:, 1::2] = b[:, 1::2].clip(0, h)
        return b.astype(np.float32)
    gt = boxes(n_gt)
    pred = boxes(n_pred)
    k = min(n_pred, n_gt)     # make some predictions near some GT so that matches are real
    pred[:k] = gt[rng.permutation(n_gt)[:k]] + rng.normal(0, 4, size=(k, 4)).astype(np.float32)
    pred_scores = rng.uniform(0.01, 0.99, size=(n_pred, 3)).astype(np.float32)
    gt_scores = rng.uniform(0.01, 0.99, size=(n_gt, 3)).astype(np.float32)
    return pred, pred_scores, gt, gt_scores


def main():
    Assigner = load_reference()
    assigner = Assigner(**ASSIGNER_CFG)
    rng = np.random.default_rng(7)
    out = dict()
    shapes = [(5, 3), (3, 5), (1, 1), (40, 17), (17, 40), (100, 100), (64, 1), (0, 4), (4, 0), (12, 12)]
    meta = dict(img_shape=(375, 1242, 3))
    for ci, (n, m) in enumerate(shapes):
        pred, ps, gt, gs = random_case(rng, n, m)
        factor = np.array([1242, 375, 1242, 375], np.float32)
        p = torch.from_numpy(pred)
        pred_norm = torch.cat([(p[:, :2] + p[:, 2:]) / 2, p[:, 2:] - p[:, :2]], dim=1) / torch.from_numpy(factor)
        pl = torch.logit(torch.from_numpy(ps), eps=1e-6)
        gl = torch.logit(torch.from_numpy(gs), eps=1e-6)
        res = assigner.assign(pred_norm, pl, torch.from_numpy(gt), gl, meta)
        out['c%d_pred_norm' % ci] = pred_norm.numpy()
        out['c%d_pred_logits' % ci] = pl.numpy()
        out['c%d_gt' % ci] = gt
        out['c%d_gt_logits' % ci] = gl.numpy()
        out['c%d_gt_inds' % ci] = res.gt_inds.numpy()
        out['c%d_has_cost' % ci] = np.array(res.max_overlaps is not None)
        if res.max_overlaps is not None:
            out['c%d_max_overlaps' % ci] = res.max_overlaps.numpy()
            out['c%d_iou_cost' % ci] = res.iou_cost.numpy()
            # the cost matrix the reference handed to scipy, recomputed the way assign() does
            cc = assigner.cls_cost(pl, gl)
            rc = assigner.reg_cost(pred_norm, torch.from_numpy(gt) / torch.from_numpy(factor))
            from detmatch_amd.mm3d.losses import bbox_cxcywh_to_xyxy
            ic = assigner.iou_cost(bbox_cxcywh_to_xyxy(pred_norm) * torch.from_numpy(factor),
                                   torch.from_numpy(gt))
            cost = (cc + rc + ic).numpy().astype(np.float32)
            r, c = linear_sum_assignment(cost)
            out['c%d_cost' % ci] = cost
            out['c%d_rows' % ci] = r.astype(np.int64)
            out['c%d_cols' % ci] = c.astype(np.int64)
    # pure LAP cases (no cost formula involved): random rectangular fp32 matrices
    for li, (n, m) in enumerate([(7, 7), (30, 11), (11, 30), (100, 100), (1, 9), (128, 96)]):
        cost = rng.normal(0, 3, size=(n, m)).astype(np.float32)
        r, c = linear_sum_assignment(cost)
        out['l%d_cost' % li] = cost
        out['l%d_rows' % li] = r.astype(np.int64)
        out['l%d_cols' % li] = c.astype(np.int64)
    out['n_cases'] = np.array(len(shapes))
    out['n_lap'] = np.array(6)
    np.savez_compressed(os.path.join(HERE, 'hungarian.npz'), **out)
    print('wrote hungarian.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
