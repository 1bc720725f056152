"""Generates tests/golden/pcdet_roi_sampler.npz by running the REFERENCE's own second-stage target
assignment end to end (thirdparty/Spconv-OpenPCDet/pcdet, loaded by file path; build container only):

    pcdet/models/roi_heads/target_assigner/proposal_target_layer.py   ProposalTargetLayer.forward,
        sample_rois_for_rcnn, get_max_iou_with_same_class, subsample_rois, sample_bg_inds  (:13-259)
    pcdet/models/roi_heads/roi_head_template.py                       RoIHeadTemplate.assign_targets (:104-134)

The one GPU-only call on that path, iou3d_nms_utils.boxes_iou3d_gpu, is served by the CPU oracle's
boxes_iou3d (oracle/dm_oracle.c — pinned bit for bit to the reference's compiled iou3d_cpu.cpp,
tests/golden/iou3d_ref.npz).  Per scene the fixture stores the inputs and

  * `all_*`: the reference's outputs with EVERY RoI "sampled" (subsample_rois patched to arange,
    ROI_PER_IMAGE = number of RoIs): the per-RoI max IoU / assigned GT, reg_valid_mask, soft class label and
    canonical GT of the whole target layer — the deterministic function each sampled row must satisfy;
  * `smp_*`: the reference's real (seeded) sample: its foreground / hard / easy background counts, which
    are deterministic functions of the candidate set sizes (the identities of the drawn rows are random in
    the reference as well).

    python tests/golden/gen_roi_sampler_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import gen_pcdet_golden as G  # noqa: E402


def scene(seed, n_rois=512, n_gt=12, mode='mixed'):
    """RoIs scattered around ground-truth boxes so that IoUs cover [0, 1] (as tests/test_roi_targets_gpu.py)."""
    g = torch.Generator().manual_seed(seed)
    B = 2
    gt = torch.zeros(B, n_gt + 3, 8)
    gt[:, :n_gt, 0] = torch.rand(B, n_gt, generator=g) * 60
    gt[:, :n_gt, 1] = torch.rand(B, n_gt, generator=g) * 60 - 30
    gt[:, :n_gt, 2] = torch.rand(B, n_gt, generator=g) - 1.5
    gt[:, :n_gt, 3:6] = torch.tensor([3.9, 1.6, 1.56]) * (0.8 + 0.4 * torch.rand(B, n_gt, 3, generator=g))
    gt[:, :n_gt, 6] = (torch.rand(B, n_gt, generator=g) - 0.5) * 7
    gt[:, :n_gt, 7] = torch.randint(1, 4, (B, n_gt), generator=g).float()
    if mode == 'mixed':
        gt[1, 5] = 0                   # a zero row in the middle stays a candidate (only the tail is padding)
    src = torch.randint(0, n_gt, (B, n_rois), generator=g)
    rois = torch.gather(gt[:, :, :7], 1, src[..., None].expand(-1, -1, 7)).clone()
    spread = {'mixed': 1.0, 'all_fg': 0.0, 'no_fg': 0.3}[mode]
    scale = torch.rand(B, n_rois, 1, generator=g) * spread
    rois[..., 0:3] += torch.randn(B, n_rois, 3, generator=g) * scale * torch.tensor([1.2, 0.6, 0.3])
    rois[..., 3:6] *= 1 + 0.15 * scale * torch.randn(B, n_rois, 3, generator=g)
    rois[..., 6] += 0.3 * scale[..., 0] * torch.randn(B, n_rois, generator=g)
    # distinct rows (the test finds a sampled row among the inputs by its bytes)
    rois[..., 0] += torch.arange(n_rois)[None, :] * 1e-4
    if mode == 'no_fg':
        rois[..., 3:6] *= 0.55
    labels = torch.gather(gt[:, :, 7], 1, src).long()
    if mode == 'mixed':
        labels[:, ::7] = (labels[:, ::7] % 3) + 1          # some RoIs carry another class
        labels[1, 10:20] = 9                               # no ground truth of this class at all
    scores = torch.rand(B, n_rois, generator=g)
    full = torch.randn(B, n_rois, 3, generator=g)
    return dict(batch_size=B, rois=rois, roi_scores=scores, roi_labels=labels, gt_boxes=gt, roi_scores_full=full)


def main():
    import oracle
    from detmatch_amd import configs
    oracle.build()
    G.load_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self          # loss_utils.py:97 (.cuda() of a constant)
    for n in ('pcdet.models.model_utils', 'pcdet.models.roi_heads', 'pcdet.models.roi_heads.target_assigner'):
        G._pkg(n)
    G._load('pcdet.models.model_utils.model_nms_utils', 'pcdet/models/model_utils/model_nms_utils.py')
    G._load('pcdet.models.roi_heads.target_assigner.proposal_target_layer',
            'pcdet/models/roi_heads/target_assigner/proposal_target_layer.py')
    tmpl = G._load('pcdet.models.roi_heads.roi_head_template', 'pcdet/models/roi_heads/roi_head_template.py')
    iou_utils = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils']
    iou_utils.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(
        oracle.boxes_iou3d(a.detach().numpy(), b.detach().numpy()))

    def to_attr(d):
        return G.AttrDict({k: (to_attr(v) if isinstance(v, dict) else v) for k, v in d.items()})
    out = {}
    # ('all_fg': the reference's own branch for "no background at all" concatenates a tensor with a list,
    # proposal_target_layer.py:159-183, and raises — not reproducible)
    cases = [('mixed', 0), ('mixed', 1), ('no_fg', 3)]
    out['n_cases'] = np.array(len(cases))
    for ci, (mode, seed) in enumerate(cases):
        k = 's%d_' % ci
        sc = scene(seed, mode=mode)
        for name in ('rois', 'roi_scores', 'roi_labels', 'gt_boxes', 'roi_scores_full'):
            out[k + name] = sc[name].numpy().copy()
        # (1) every RoI "sampled"
        cfg = to_attr(configs.pvrcnn_kitti_model()['pcdet_model']['ROI_HEAD'])
        cfg.TARGET_CONFIG['ROI_PER_IMAGE'] = sc['rois'].shape[1]
        head = tmpl.RoIHeadTemplate(num_class=1, model_cfg=cfg)
        head.proposal_target_layer.subsample_rois = lambda max_overlaps: torch.arange(len(max_overlaps))
        td = head.assign_targets({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in sc.items()})
        out.update({k + 'all_iou': td['gt_iou_of_rois'].numpy(), k + 'all_reg_valid': td['reg_valid_mask'].numpy(),
                    k + 'all_cls_labels': td['rcnn_cls_labels'].numpy(), k + 'all_gt_of_rois': td['gt_of_rois'].numpy(),
                    k + 'all_gt_of_rois_src': td['gt_of_rois_src'].numpy()})
        # (2) the reference's real sample
        cfg2 = to_attr(configs.pvrcnn_kitti_model()['pcdet_model']['ROI_HEAD'])
        head2 = tmpl.RoIHeadTemplate(num_class=1, model_cfg=cfg2)
        np.random.seed(seed)
        torch.manual_seed(seed)
        ts = head2.assign_targets({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in sc.items()})
        iou = ts['gt_iou_of_rois'].numpy()
        t = cfg2.TARGET_CONFIG
        fg_thr = min(t['REG_FG_THRESH'], t['CLS_FG_THRESH'])
        out[k + 'smp_n_fg'] = (iou >= fg_thr).sum(1)
        out[k + 'smp_n_hard'] = ((iou < t['REG_FG_THRESH']) & (iou >= t['CLS_BG_THRESH_LO'])).sum(1)
        out[k + 'smp_n_easy'] = (iou < t['CLS_BG_THRESH_LO']).sum(1)
        out[k + 'smp_rois'] = ts['rois'].numpy()
    np.savez_compressed(os.path.join(HERE, 'pcdet_roi_sampler.npz'), **out)
    print('wrote pcdet_roi_sampler.npz:', {c: (out['s%d_smp_n_fg' % i].tolist(), out['s%d_smp_n_hard' % i].tolist(),
                                                out['s%d_smp_n_easy' % i].tolist()) for i, c in enumerate(cases)})


if __name__ == '__main__':
    main()
