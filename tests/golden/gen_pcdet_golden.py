"""Generates tests/golden/pcdet_torch.npz by running the REFERENCE's own pure-PyTorch OpenPCDet code
(thirdparty/Spconv-OpenPCDet/pcdet, loaded by file path; build container only) on seeded inputs:

    pcdet/utils/box_coder_utils.py     ResidualCoder.encode_torch / decode_torch
    pcdet/utils/loss_utils.py          SigmoidFocalClassificationLoss, WeightedSmoothL1Loss,
                                       WeightedCrossEntropyLoss, get_corner_loss_lidar
    pcdet/utils/box_utils.py           boxes_to_corners_3d, boxes3d_nearest_bev_iou, enlarge_box3d
    pcdet/utils/common_utils.py        limit_period, rotate_points_along_z, get_voxel_centers
    pcdet/models/dense_heads/target_assigner/anchor_generator.py          AnchorGenerator
    pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py  assign_targets

The compiled extension modules those files import (iou3d_nms_cuda, roiaware_pool3d_cuda) are not
built here and not needed on these code paths: empty placeholder modules stand in for them, and
Tensor.cuda() is made the identity for the two `.cuda()` calls of the anchor generator.  The fixture
holds inputs + reference outputs only.

    python tests/golden/gen_pcdet_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/thirdparty/Spconv-OpenPCDet'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(dotted, rel):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    parent, _, leaf = dotted.rpartition('.')
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, mod)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for n in ('pcdet', 'pcdet.ops', 'pcdet.ops.roiaware_pool3d', 'pcdet.ops.iou3d_nms', 'pcdet.utils',
              'pcdet.models', 'pcdet.models.dense_heads', 'pcdet.models.dense_heads.target_assigner'):
        _pkg(n)
    sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda'] = types.ModuleType('roiaware_pool3d_cuda')
    sys.modules['pcdet.ops.roiaware_pool3d'].roiaware_pool3d_cuda = sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda']
    sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_cuda'] = types.ModuleType('iou3d_nms_cuda')
    sys.modules['pcdet.ops.iou3d_nms'].iou3d_nms_cuda = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_cuda']
    common = _load('pcdet.utils.common_utils', 'pcdet/utils/common_utils.py')
    _load('pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils', 'pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py')
    box_utils = _load('pcdet.utils.box_utils', 'pcdet/utils/box_utils.py')
    coder = _load('pcdet.utils.box_coder_utils', 'pcdet/utils/box_coder_utils.py')
    loss = _load('pcdet.utils.loss_utils', 'pcdet/utils/loss_utils.py')
    _load('pcdet.ops.iou3d_nms.iou3d_nms_utils', 'pcdet/ops/iou3d_nms/iou3d_nms_utils.py')
    agen = _load('pcdet.models.dense_heads.target_assigner.anchor_generator',
                 'pcdet/models/dense_heads/target_assigner/anchor_generator.py')
    assigner = _load('pcdet.models.dense_heads.target_assigner.axis_aligned_target_assigner',
                     'pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py')
    return common, box_utils, coder, loss, agen, assigner


class AttrDict(dict):
    __getattr__ = dict.__getitem__

    def get(self, k, d=None):
        return dict.get(self, k, d)


def main():
    from detmatch_amd import configs, synth
    common, box_utils, coder_m, loss_m, agen_m, assigner_m = load_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self          # anchor_generator.py:36,39
    rng = np.random.default_rng(11)
    out = {}
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))

    def boxes(n):
        return np.concatenate([rng.uniform([0, -40, -3], [70, 40, 1], (n, 3)), rng.uniform(0.4, 5.0, (n, 3)),
                               rng.uniform(-4, 4, (n, 1))], 1).astype(np.float32)
    # ---- common_utils
    ang = rng.uniform(-7, 7, 50).astype(np.float32)
    out['lp_in'] = ang
    out['lp_out_pi'] = common.limit_period(f32(ang), 0.5, np.pi).numpy()
    out['lp_out_2pi'] = common.limit_period(f32(ang), 0.5, 2 * np.pi).numpy()
    pts = rng.normal(size=(4, 9, 5)).astype(np.float32)
    rot = rng.uniform(-3, 3, 4).astype(np.float32)
    out['rot_pts'], out['rot_ang'] = pts, rot
    out['rot_out'] = common.rotate_points_along_z(f32(pts), f32(rot)).numpy()
    vc = rng.integers(0, 40, (30, 3)).astype(np.int32)
    out['vc_in'] = vc
    out['vc_out'] = common.get_voxel_centers(torch.from_numpy(vc), 4, [0.05, 0.05, 0.1],
                                             [0, -40, -3, 70.4, 40, 1]).numpy()
    # ---- box_utils
    b = boxes(40)
    out['bx_in'] = b
    out['bx_corners'] = box_utils.boxes_to_corners_3d(f32(b)).numpy()
    out['bx_enlarged'] = box_utils.enlarge_box3d(f32(b), extra_width=(0.2, 0.2, 0.2)).numpy()
    b2 = boxes(17)
    b2[:5] = b[:5] + rng.normal(0, 0.2, (5, 7)).astype(np.float32)
    out['bx_in2'] = b2
    out['bx_nearest_iou'] = box_utils.boxes3d_nearest_bev_iou(f32(b), f32(b2)).numpy()
    # ---- ResidualCoder
    rc = coder_m.ResidualCoder()
    anchors = boxes(60)
    gts = anchors + rng.normal(0, 0.3, anchors.shape).astype(np.float32)
    gts[:, 3:6] = np.abs(gts[:, 3:6]) + 0.1
    out['rc_anchors'], out['rc_gt'] = anchors, gts
    enc = rc.encode_torch(f32(gts), f32(anchors))
    out['rc_enc'] = enc.numpy()
    out['rc_dec'] = rc.decode_torch(enc, f32(anchors)).numpy()
    # ---- losses
    logits = rng.normal(0, 2, (2, 500, 3)).astype(np.float32)
    onehot = np.eye(4, dtype=np.float32)[rng.integers(0, 4, (2, 500))][..., 1:]
    w = rng.uniform(0, 1, (2, 500)).astype(np.float32)
    out['fl_logits'], out['fl_targets'], out['fl_w'] = logits, onehot, w
    out['fl_out'] = loss_m.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(f32(logits), f32(onehot), f32(w)).numpy()
    pr, tg = rng.normal(0, 1, (2, 300, 7)).astype(np.float32), rng.normal(0, 1, (2, 300, 7)).astype(np.float32)
    tg[0, :5, 2] = np.nan
    cw = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
    out['sl_pred'], out['sl_tgt'], out['sl_w'] = pr, tg, w[:, :300]
    out['sl_out'] = loss_m.WeightedSmoothL1Loss(beta=1.0 / 9.0, code_weights=cw)(f32(pr), f32(tg), f32(w[:, :300])).numpy()
    dl = rng.normal(0, 1, (2, 300, 2)).astype(np.float32)
    dt = np.eye(2, dtype=np.float32)[rng.integers(0, 2, (2, 300))]
    out['ce_logits'], out['ce_tgt'] = dl, dt
    out['ce_out'] = loss_m.WeightedCrossEntropyLoss()(f32(dl), f32(dt), f32(w[:, :300])).numpy()
    pb, gb = boxes(64), boxes(64)
    out['cl_pred'], out['cl_gt'] = pb, gb
    out['cl_out'] = loss_m.get_corner_loss_lidar(f32(pb), f32(gb)).numpy()
    # ---- anchors + target assignment (KITTI PV-RCNN config)
    cfg = configs.pvrcnn_kitti_model()['pcdet_model']['DENSE_HEAD']
    gen_cfg = [AttrDict(c) for c in cfg['ANCHOR_GENERATOR_CONFIG']]
    gen = agen_m.AnchorGenerator(anchor_range=np.array(configs.POINT_CLOUD_RANGE, dtype=np.float32),
                                 anchor_generator_config=gen_cfg)
    fmap = [np.array([176, 200]) for _ in gen_cfg]
    anchors_list, per_loc = gen.generate_anchors(fmap)
    out['ag_num_per_loc'] = np.array(per_loc)
    for i, a in enumerate(anchors_list):
        out['ag_shape_%d' % i] = np.array(a.shape)
        out['ag_sample_%d' % i] = a.reshape(-1, 7)[::997].numpy()      # every 997th anchor
        out['ag_sum_%d' % i] = a.double().sum(dim=(0, 1, 2, 3, 4)).numpy()
    model_cfg = AttrDict(ANCHOR_GENERATOR_CONFIG=gen_cfg,
                         TARGET_ASSIGNER_CONFIG=AttrDict(cfg['TARGET_ASSIGNER_CONFIG']))
    ta = assigner_m.AxisAlignedTargetAssigner(model_cfg=model_cfg, class_names=configs.CLASS_NAMES,
                                              box_coder=rc, match_height=False)
    gt = np.zeros((3, 12, 8), np.float32)
    for s in range(2):
        f = synth.lidar_frame(s)
        lab = synth._SIM_TO_CFG_LABEL[f['gt_labels']] + 1
        g = np.concatenate([f['gt_boxes'], lab[:, None].astype(np.float32)], 1)
        if s == 1:
            g = g[:7]
        gt[s, :len(g)] = g
    out['ta_gt'] = gt                      # third sample: no GT at all
    res = ta.assign_targets(anchors_list, f32(gt))
    out['ta_labels'] = res['box_cls_labels'].numpy().astype(np.int32)
    out['ta_weights'] = res['reg_weights'].numpy()
    fg = out['ta_labels'] > 0
    out['ta_fg_idx'] = np.stack(np.nonzero(fg), 1).astype(np.int32)
    out['ta_fg_targets'] = res['box_reg_targets'].numpy()[fg]
    np.savez_compressed(os.path.join(HERE, 'pcdet_torch.npz'), **out)
    print('wrote pcdet_torch.npz: %d arrays, %d foreground anchors' % (len(out), int(fg.sum())))
    roi_head_goldens(rng, boxes)
    dense_goldens(rng, rc)
    state_key_goldens()


def roi_head_goldens(rng, boxes):
    """RoIHeadTemplate (roi_head_template.py) with the sampling step replaced by fixed inputs: the
    IoU -> (reg_valid_mask, soft cls label) mapping of ProposalTargetLayer.forward, the canonical
    transform of assign_targets, generate_predicted_boxes and the two RoI losses."""
    from detmatch_amd import configs
    for n in ('pcdet.models.model_utils', 'pcdet.models.roi_heads', 'pcdet.models.roi_heads.target_assigner'):
        _pkg(n)
    _load('pcdet.models.model_utils.model_nms_utils', 'pcdet/models/model_utils/model_nms_utils.py')
    ptl_m = _load('pcdet.models.roi_heads.target_assigner.proposal_target_layer',
                  'pcdet/models/roi_heads/target_assigner/proposal_target_layer.py')
    tmpl = _load('pcdet.models.roi_heads.roi_head_template', 'pcdet/models/roi_heads/roi_head_template.py')

    def to_attr(d):
        return AttrDict({k: (to_attr(v) if isinstance(v, dict) else v) for k, v in d.items()})
    cfg = to_attr(configs.pvrcnn_kitti_model()['pcdet_model']['ROI_HEAD'])
    head = tmpl.RoIHeadTemplate(num_class=1, model_cfg=cfg)
    f32 = lambda a: torch.from_numpy(np.array(a, dtype=np.float32, copy=True))   # the reference edits in place
    B, R = 2, 128
    rois = boxes(B * R).reshape(B, R, 7)
    gt = rois + rng.normal(0, 0.4, rois.shape).astype(np.float32)
    gt[..., 3:6] = np.abs(gt[..., 3:6]) + 0.2
    gt_of_rois = np.concatenate([gt, rng.integers(1, 4, (B, R, 1)).astype(np.float32)], -1)
    ious = rng.uniform(0, 1, (B, R)).astype(np.float32)
    scores = rng.normal(size=(B, R)).astype(np.float32)
    labels = rng.integers(1, 4, (B, R)).astype(np.int64)
    scores_full = rng.normal(size=(B, R, 3)).astype(np.float32)
    head.proposal_target_layer.sample_rois_for_rcnn = lambda batch_dict: (
        f32(rois), f32(gt_of_rois), f32(ious), f32(scores), torch.from_numpy(labels), f32(scores_full))
    td = head.assign_targets(dict(batch_size=B))
    out = dict(rois=rois, gt_of_rois_in=gt_of_rois, ious=ious, scores=scores, labels=labels,
               scores_full=scores_full, reg_valid_mask=td['reg_valid_mask'].numpy(),
               rcnn_cls_labels=td['rcnn_cls_labels'].numpy(), gt_of_rois=td['gt_of_rois'].numpy(),
               gt_of_rois_src=td['gt_of_rois_src'].numpy())
    rcnn_cls = rng.normal(size=(B * R, 1)).astype(np.float32)
    rcnn_reg = rng.normal(0, 0.3, (B * R, 7)).astype(np.float32)
    cls_p, box_p = head.generate_predicted_boxes(B, f32(rois), f32(rcnn_cls), f32(rcnn_reg))
    out.update(rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg, pred_cls=cls_p.numpy(), pred_boxes=box_p.numpy())
    fwd = dict(td)
    fwd['rcnn_cls'], fwd['rcnn_reg'] = f32(rcnn_cls), f32(rcnn_reg)
    head.forward_ret_dict = fwd
    lc, _ = head.get_box_cls_layer_loss(fwd)
    lr, tb = head.get_box_reg_layer_loss(fwd)
    out.update(loss_cls=np.array(float(lc)), loss_reg=np.array(float(lr)),
               loss_corner=np.array(float(tb['rcnn_loss_corner'])))
    # a batch without any regression-valid RoI
    fwd0 = dict(fwd)
    fwd0['reg_valid_mask'] = torch.zeros_like(fwd['reg_valid_mask'])
    lr0, _ = head.get_box_reg_layer_loss(fwd0)
    out['loss_reg_no_fg'] = np.array(float(lr0))
    np.savez_compressed(os.path.join(HERE, 'pcdet_roi_head.npz'), **out)
    print('wrote pcdet_roi_head.npz: %d arrays; %d reg-valid RoIs' % (len(out), int(out['reg_valid_mask'].sum())))




def _to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: _to_attr(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return type(d)(_to_attr(v) for v in d)
    return d


def dense_goldens(rng, rc):
    """Module-level goldens (state_dict + input -> output, losses) of the reference's dense modules on a
    REDUCED geometry (16 m x 16 m range, narrow channels) so that the fixture stays small:
    AnchorHeadSingle forward/get_loss/generate_predicted_boxes, BaseBEVBackbone.forward,
    PointHeadSimple forward/get_loss, MeanVFE.forward, bilinear_interpolate_torch."""
    from detmatch_amd import configs
    for n in ('pcdet.models.backbones_2d', 'pcdet.models.backbones_3d', 'pcdet.models.backbones_3d.vfe',
              'pcdet.models.backbones_3d.pfe', 'pcdet.ops.pointnet2', 'pcdet.ops.pointnet2.pointnet2_stack'):
        _pkg(n)
    for leaf in ('pointnet2_modules', 'pointnet2_utils'):     # imported by voxel_set_abstraction.py, unused here
        m = types.ModuleType('pcdet.ops.pointnet2.pointnet2_stack.' + leaf)
        sys.modules[m.__name__] = m
        setattr(sys.modules['pcdet.ops.pointnet2.pointnet2_stack'], leaf, m)
    _load('pcdet.models.dense_heads.target_assigner.atss_target_assigner',
          'pcdet/models/dense_heads/target_assigner/atss_target_assigner.py')
    _load('pcdet.models.dense_heads.anchor_head_template', 'pcdet/models/dense_heads/anchor_head_template.py')
    ahs = _load('pcdet.models.dense_heads.anchor_head_single', 'pcdet/models/dense_heads/anchor_head_single.py')
    _load('pcdet.models.dense_heads.point_head_template', 'pcdet/models/dense_heads/point_head_template.py')
    phs = _load('pcdet.models.dense_heads.point_head_simple', 'pcdet/models/dense_heads/point_head_simple.py')
    _load('pcdet.utils.frozen_bn', 'pcdet/utils/frozen_bn.py')      # imported, not selected (no FROZEN_BN key)
    bev = _load('pcdet.models.backbones_2d.base_bev_backbone', 'pcdet/models/backbones_2d/base_bev_backbone.py')
    _load('pcdet.models.backbones_3d.vfe.vfe_template', 'pcdet/models/backbones_3d/vfe/vfe_template.py')
    vfe = _load('pcdet.models.backbones_3d.vfe.mean_vfe', 'pcdet/models/backbones_3d/vfe/mean_vfe.py')
    vsa = _load('pcdet.models.backbones_3d.pfe.voxel_set_abstraction',
                'pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py')
    model = configs.pvrcnn_kitti_model()['pcdet_model']
    out = {}
    f32 = lambda a: torch.from_numpy(np.array(a, dtype=np.float32, copy=True))
    sd = lambda prefix, m: out.update({prefix + '/' + k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
    torch.manual_seed(5)
    # ---- AnchorHeadSingle on a 40 x 40 map
    pcr = np.array([0, -8, -3, 16, 8, 1], dtype=np.float32)
    grid = np.array([320, 320, 40])
    C = 16
    head = ahs.AnchorHeadSingle(model_cfg=_to_attr(model['DENSE_HEAD']), input_channels=C, num_class=3,
                                class_names=configs.CLASS_NAMES, grid_size=grid, point_cloud_range=pcr)
    with torch.no_grad():
        for p in head.parameters():                      # conv_box is N(0, 0.001): make the maps non-trivial
            p.add_(torch.randn_like(p) * 0.1)
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)
    z0 = np.array([-1.0, -0.6, -0.6], np.float32)
    gt = np.zeros((2, 9, 8), np.float32)
    for b, n in enumerate((9, 4)):
        cls = rng.integers(0, 3, n)
        gt[b, :n, 0] = rng.uniform(1, 15, n)
        gt[b, :n, 1] = rng.uniform(-7, 7, n)
        gt[b, :n, 2] = z0[cls] + rng.normal(0, 0.1, n)
        gt[b, :n, 3:6] = sizes[cls] * rng.uniform(0.9, 1.1, (n, 3))
        gt[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        gt[b, :n, 7] = cls + 1
    x = rng.normal(size=(2, C, 40, 40)).astype(np.float32)
    head.train()
    dd = head(dict(spatial_features_2d=f32(x), gt_boxes=f32(gt), batch_size=2))
    sd('ah', head)
    out.update(ah_pcr=pcr, ah_grid=grid, ah_x=x, ah_gt=gt,
               ah_cls_preds=head.forward_ret_dict['cls_preds'].detach().numpy(),
               ah_box_preds=head.forward_ret_dict['box_preds'].detach().numpy(),
               ah_dir_preds=head.forward_ret_dict['dir_cls_preds'].detach().numpy(),
               ah_labels=head.forward_ret_dict['box_cls_labels'].numpy().astype(np.int32),
               ah_batch_cls=dd['batch_cls_preds'].detach().numpy(),
               ah_batch_box=dd['batch_box_preds'].detach().numpy())
    loss, tb = head.get_loss()
    out.update(ah_loss=np.array(float(loss)), ah_loss_cls=np.array(tb['rpn_loss_cls']),
               ah_loss_loc=np.array(tb['rpn_loss_loc']), ah_loss_dir=np.array(tb['rpn_loss_dir']))
    # ---- BaseBEVBackbone, narrow
    bcfg = AttrDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16], UPSAMPLE_STRIDES=[1, 2],
                    NUM_UPSAMPLE_FILTERS=[16, 16])
    bb = bev.BaseBEVBackbone(bcfg, input_channels=12)
    bb.train()                                            # batch statistics: exercises eps / momentum too
    xb = rng.normal(size=(2, 12, 24, 20)).astype(np.float32)
    sd('bev_before', bb)
    ob = bb(dict(spatial_features=f32(xb)))
    sd('bev_after', bb)
    out.update(bev_x=xb, bev_out=ob['spatial_features_2d'].detach().numpy())
    # ---- PointHeadSimple
    ph = phs.PointHeadSimple(num_class=1, input_channels=24, model_cfg=_to_attr(model['POINT_HEAD']))
    ph.eval()
    with torch.no_grad():
        for m in ph.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    pf = rng.normal(size=(300, 24)).astype(np.float32)
    od = ph(dict(point_features=f32(pf), point_features_before_fusion=f32(pf),
                 point_coords=torch.zeros(300, 4), batch_size=1))
    sd('ph', ph)
    lab = rng.integers(-1, 2, 300).astype(np.int64)
    ph.forward_ret_dict['point_cls_labels'] = torch.from_numpy(lab)
    pl, ptb = ph.get_loss()
    out.update(ph_x=pf, ph_scores=od['point_cls_scores'].detach().numpy(),
               ph_preds=ph.forward_ret_dict['point_cls_preds'].detach().numpy(), ph_labels=lab,
               ph_loss=np.array(float(pl)), ph_pos=np.array(ptb['point_pos_num']))
    # ---- MeanVFE
    mv = vfe.MeanVFE(model_cfg=AttrDict(), num_point_features=4)
    vox = rng.normal(size=(50, 5, 4)).astype(np.float32)
    npts = rng.integers(1, 6, 50).astype(np.int32)
    for i, n in enumerate(npts):
        vox[i, n:] = 0
    out.update(vfe_voxels=vox, vfe_num=npts,
               vfe_out=mv(dict(voxels=f32(vox), voxel_num_points=torch.from_numpy(npts)))['voxel_features'].numpy())
    # ---- bilinear interpolation of BEV features at keypoints
    im = rng.normal(size=(25, 22, 6)).astype(np.float32)
    bx = rng.uniform(-1, 23, 80).astype(np.float32)      # includes out-of-map coordinates (clamped)
    by = rng.uniform(-1, 26, 80).astype(np.float32)
    out.update(bi_im=im, bi_x=bx, bi_y=by,
               bi_out=vsa.bilinear_interpolate_torch(f32(im), f32(bx), f32(by)).numpy())
    # ---- SparseConvTensor.dense + HeightCompression (structure.py has no compiled import)
    for n in ('pcdet.ops.spconv', 'pcdet.models.backbones_2d.map_to_bev'):
        if n not in sys.modules:
            _pkg(n)
    st = _load('pcdet.ops.spconv.structure', 'pcdet/ops/spconv/structure.py')
    hc = _load('pcdet.models.backbones_2d.map_to_bev.height_compression',
               'pcdet/models/backbones_2d/map_to_bev/height_compression.py')
    cells = rng.permutation(2 * 2 * 5 * 7)[:50]
    sp_idx = np.stack(np.unravel_index(cells, (2, 2, 5, 7)), 1).astype(np.int32)
    sp_feat = rng.normal(size=(50, 6)).astype(np.float32)
    spt = st.SparseConvTensor(f32(sp_feat), torch.from_numpy(sp_idx.copy()), [2, 5, 7], 2)
    hd = hc.HeightCompression(AttrDict(NUM_BEV_FEATURES=12))(dict(encoded_spconv_tensor=spt,
                                                                 encoded_spconv_tensor_stride=8))
    out.update(sp_idx=sp_idx, sp_feat=sp_feat, sp_dense=spt.dense().numpy(),
               sp_bev=hd['spatial_features'].numpy())
    np.savez_compressed(os.path.join(HERE, 'pcdet_dense.npz'), **out)
    print('wrote pcdet_dense.npz: %d arrays, %d positive anchors, %.0f kB' % (
        len(out), int((out['ah_labels'] > 0).sum()), os.path.getsize(os.path.join(HERE, 'pcdet_dense.npz')) / 1e3))


def state_key_goldens():
    """Parameter / buffer names and shapes of the reference's PV-RCNN modules at the KITTI config
    (modules are only constructed; the compiled ops behind them are placeholder modules): the
    checkpoint-interoperability contract of the 3D branch."""
    import json
    from detmatch_amd import configs
    for n in ('pcdet.ops.spconv',):
        _pkg(n)
    sys.modules['pcdet.ops.spconv.sparse_conv_ext'] = types.ModuleType('sparse_conv_ext')
    sys.modules['pcdet.ops.spconv'].sparse_conv_ext = sys.modules['pcdet.ops.spconv.sparse_conv_ext']
    for leaf in ('structure', 'modules', 'ops', 'functional', 'conv', 'pool'):
        _load('pcdet.ops.spconv.' + leaf, 'pcdet/ops/spconv/%s.py' % leaf)
    sp = _load('pcdet.ops.spconv', 'pcdet/ops/spconv/__init__.py')
    sp.__path__ = []
    sys.modules['pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda'] = types.ModuleType('pointnet2_stack_cuda')
    sys.modules['pcdet.ops.pointnet2.pointnet2_stack'].pointnet2_stack_cuda = \
        sys.modules['pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda']
    _load('pcdet.ops.pointnet2.pointnet2_stack.pointnet2_utils', 'pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py')
    _load('pcdet.ops.pointnet2.pointnet2_stack.pointnet2_modules', 'pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py')
    bb3 = _load('pcdet.models.backbones_3d.spconv_backbone', 'pcdet/models/backbones_3d/spconv_backbone.py')
    vsa = _load('pcdet.models.backbones_3d.pfe.voxel_set_abstraction',
                'pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py')
    pvh = _load('pcdet.models.roi_heads.pvrcnn_head', 'pcdet/models/roi_heads/pvrcnn_head.py')
    m = sys.modules
    model = _to_attr(configs.pvrcnn_kitti_model()['pcdet_model'])
    grid = np.array([1408, 1600, 40])
    pcr = np.array(configs.POINT_CLOUD_RANGE, dtype=np.float32)
    mods = {
        'backbone_3d': bb3.VoxelBackBone8x(model.BACKBONE_3D, input_channels=4, grid_size=grid),
        'pfe': vsa.VoxelSetAbstraction(model.PFE, voxel_size=[0.05, 0.05, 0.1], point_cloud_range=pcr,
                                       num_bev_features=256, num_rawpoint_features=4),
        'backbone_2d': m['pcdet.models.backbones_2d.base_bev_backbone'].BaseBEVBackbone(model.BACKBONE_2D, 256),
        'dense_head': m['pcdet.models.dense_heads.anchor_head_single'].AnchorHeadSingle(
            model_cfg=model.DENSE_HEAD, input_channels=512, num_class=3, class_names=configs.CLASS_NAMES,
            grid_size=grid, point_cloud_range=pcr),
    }
    mods['point_head'] = m['pcdet.models.dense_heads.point_head_simple'].PointHeadSimple(
        num_class=1 if model.POINT_HEAD.CLASS_AGNOSTIC else 3,
        input_channels=(mods['pfe'].num_point_features_before_fusion
                        if model.POINT_HEAD.get('USE_POINT_FEATURES_BEFORE_FUSION', False)
                        else mods['pfe'].num_point_features),     # detector3d_template.py:143-146
        model_cfg=model.POINT_HEAD)
    mods['roi_head'] = pvh.PVRCNNHead(input_channels=mods['pfe'].num_point_features, model_cfg=model.ROI_HEAD,
                                      num_class=1 if model.ROI_HEAD.CLASS_AGNOSTIC else 3)
    keys = {}
    for name, mod in mods.items():
        for k, v in mod.state_dict().items():
            keys['%s.%s' % (name, k)] = list(v.shape)
    with open(os.path.join(HERE, 'pcdet_state_keys.json'), 'w') as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print('wrote pcdet_state_keys.json: %d entries' % len(keys))
    # pure-tensor methods of the two modules that only exist as instances
    rng = np.random.default_rng(23)
    rois = np.concatenate([rng.uniform([0, -40, -3], [70, 40, 1], (20, 3)), rng.uniform(0.4, 5.0, (20, 3)),
                           rng.uniform(-4, 4, (20, 1))], 1).astype(np.float32)
    glob, local = mods['roi_head'].get_global_grid_points_of_roi(torch.from_numpy(rois.copy()), grid_size=6)
    kp = rng.uniform([0, -40, -3], [70.4, 40, 1], (2, 50, 3)).astype(np.float32)
    bevf = rng.normal(size=(2, 8, 200, 176)).astype(np.float16).astype(np.float32)   # stored as f16: exact
    pb = mods['pfe'].interpolate_from_bev_features(torch.from_numpy(kp.copy()), torch.from_numpy(bevf.copy()), 2, 8)
    np.savez_compressed(os.path.join(HERE, 'pcdet_grid.npz'), rois=rois, grid_global=glob.numpy(),
                        grid_local=local.numpy(), kp=kp, bev=bevf.astype(np.float16), kp_bev=pb.numpy())
    print('wrote pcdet_grid.npz')


if __name__ == '__main__':
    main()
