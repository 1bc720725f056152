"""The model sections of configs/detmatch/* as Python builders (same keys and values as
the reference config files, e.g. configs/detmatch/001/pretrain_pvrcnn/split_0.py:22-199),
for use where /root/reference is absent (the GPU box).  `mm3d.config.Config.fromfile` can
still load the reference's own config files unchanged when they are available."""

CLASS_NAMES = ['Pedestrian', 'Cyclist', 'Car']
POINT_CLOUD_RANGE = [0, -40, -3, 70.4, 40, 1]
VOXEL_SIZE = [0.05, 0.05, 0.1]


def _anchor(name, size, bottom, matched, unmatched):
    return dict(class_name=name, anchor_sizes=[size], anchor_rotations=[0, 1.57],
                anchor_bottom_heights=[bottom], align_center=False, feature_map_stride=8,
                matched_threshold=matched, unmatched_threshold=unmatched)


def _sa(factor, mlp, radii, nsample):
    d = dict(MLPS=[[mlp, mlp], [mlp, mlp]], POOL_RADIUS=radii, NSAMPLE=nsample)
    if factor is not None:
        d['DOWNSAMPLE_FACTOR'] = factor
    return d


def pvrcnn_kitti_model(class_names=CLASS_NAMES, point_cloud_range=POINT_CLOUD_RANGE,
                       voxel_size=VOXEL_SIZE, max_voxels=(16000, 40000)):
    """model = dict(type='OpenPCDetDetector', ...) of pretrain_pvrcnn/split_0.py:27-199."""
    return dict(
        type='OpenPCDetDetector',
        dataset_fields=dict(class_names=class_names,
                            point_feature_encoder=dict(num_point_features=4),
                            point_cloud_range=point_cloud_range, voxel_size=voxel_size,
                            depth_downsample_factor=None),
        voxel_layer=dict(max_num_points=5, point_cloud_range=point_cloud_range,
                         voxel_size=voxel_size, max_voxels=max_voxels),
        pcdet_model=dict(
            NAME='PVRCNN',
            VFE=dict(NAME='MeanVFE'),
            BACKBONE_3D=dict(NAME='VoxelBackBone8x'),
            MAP_TO_BEV=dict(NAME='HeightCompression', NUM_BEV_FEATURES=256),
            BACKBONE_2D=dict(NAME='BaseBEVBackbone', LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2],
                             NUM_FILTERS=[128, 256], UPSAMPLE_STRIDES=[1, 2],
                             NUM_UPSAMPLE_FILTERS=[256, 256]),
            DENSE_HEAD=dict(
                NAME='AnchorHeadSingle', CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True,
                DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0, NUM_DIR_BINS=2,
                ANCHOR_GENERATOR_CONFIG=[
                    _anchor('Pedestrian', [0.8, 0.6, 1.73], -0.6, 0.5, 0.35),
                    _anchor('Cyclist', [1.76, 0.6, 1.73], -0.6, 0.5, 0.35),
                    _anchor('Car', [3.9, 1.6, 1.56], -1.78, 0.6, 0.45)],
                TARGET_ASSIGNER_CONFIG=dict(NAME='AxisAlignedTargetAssigner', POS_FRACTION=-1,
                                            SAMPLE_SIZE=512, NORM_BY_NUM_EXAMPLES=False,
                                            MATCH_HEIGHT=False, BOX_CODER='ResidualCoder'),
                LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1, loc_weight=2, dir_weight=0.2,
                                                   code_weights=[1, 1, 1, 1, 1, 1, 1]))),
            PFE=dict(
                NAME='VoxelSetAbstraction', POINT_SOURCE='raw_points', NUM_KEYPOINTS=2048,
                NUM_OUTPUT_FEATURES=128, SAMPLE_METHOD='FPS',
                FEATURES_SOURCE=['bev', 'x_conv1', 'x_conv2', 'x_conv3', 'x_conv4', 'raw_points'],
                SA_LAYER=dict(raw_points=_sa(None, 16, [0.4, 0.8], [16, 16]),
                              x_conv1=_sa(1, 16, [0.4, 0.8], [16, 16]),
                              x_conv2=_sa(2, 32, [0.8, 1.2], [16, 32]),
                              x_conv3=_sa(4, 64, [1.2, 2.4], [16, 32]),
                              x_conv4=_sa(8, 64, [2.4, 4.8], [16, 32]))),
            POINT_HEAD=dict(NAME='PointHeadSimple', CLS_FC=[256, 256], CLASS_AGNOSTIC=True,
                            USE_POINT_FEATURES_BEFORE_FUSION=True,
                            TARGET_CONFIG=dict(GT_EXTRA_WIDTH=[0.2, 0.2, 0.2]),
                            LOSS_CONFIG=dict(LOSS_REG='smooth-l1',
                                             LOSS_WEIGHTS=dict(point_cls_weight=1))),
            ROI_HEAD=dict(
                NAME='PVRCNNHead', CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256],
                REG_FC=[256, 256], DP_RATIO=0.3,
                NMS_CONFIG=dict(
                    TRAIN=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=9000,
                               NMS_POST_MAXSIZE=512, NMS_THRESH=0.8),
                    TEST=dict(NMS_TYPE='nms_gpu', MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1024,
                              NMS_POST_MAXSIZE=100, NMS_THRESH=0.7)),
                ROI_GRID_POOL=dict(GRID_SIZE=6, MLPS=[[64, 64], [64, 64]], POOL_RADIUS=[0.8, 1.6],
                                   NSAMPLE=[16, 16], POOL_METHOD='max_pool'),
                TARGET_CONFIG=dict(BOX_CODER='ResidualCoder', ROI_PER_IMAGE=128, FG_RATIO=0.5,
                                   SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE='roi_iou',
                                   CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1,
                                   HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55),
                LOSS_CONFIG=dict(CLS_LOSS='BinaryCrossEntropy', REG_LOSS='smooth-l1',
                                 CORNER_LOSS_REGULARIZATION=True,
                                 LOSS_WEIGHTS=dict(rcnn_cls_weight=1, rcnn_reg_weight=1,
                                                   rcnn_corner_weight=1,
                                                   code_weights=[1, 1, 1, 1, 1, 1, 1]))),
            POST_PROCESSING=dict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], SCORE_THRESH=0.1,
                                 OUTPUT_RAW_SCORE=False, EVAL_METRIC='kitti',
                                 NMS_CONFIG=dict(MULTI_CLASSES_NMS=False, NMS_TYPE='nms_gpu',
                                                 NMS_THRESH=0.1, NMS_PRE_MAXSIZE=4096,
                                                 NMS_POST_MAXSIZE=500))))


# ---------------------------------------------------------------------------------------------
# configs/detmatch/001/detmatch/split_0.py as builders (values only)
# ---------------------------------------------------------------------------------------------
def frcnn_kitti_model(num_classes=3):
    """detector_2d of split_0.py:39-99."""
    coder = lambda stds: dict(type='DeltaXYWHBBoxCoder', target_means=[0.0] * 4, target_stds=stds)
    return dict(
        type='FasterRCNN',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe',
                      init_cfg=dict(type='Pretrained',
                                    checkpoint='open-mmlab://detectron2/resnet50_caffe')),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256,
                      anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                            strides=[4, 8, 16, 32, 64]),
                      bbox_coder=coder([1.0, 1.0, 1.0, 1.0]),
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
        roi_head=dict(type='StandardRoIHead',
                      bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                              roi_layer=dict(type='RoIAlign', output_size=7,
                                                             sampling_ratio=0),
                                              out_channels=256, featmap_strides=[4, 8, 16, 32]),
                      bbox_head=dict(type='Shared2FCBBoxHead', in_channels=256, fc_out_channels=1024,
                                     roi_feat_size=7, num_classes=num_classes,
                                     bbox_coder=coder([0.1, 0.1, 0.2, 0.2]), reg_class_agnostic=False,
                                     loss_cls=dict(type='FocalLoss', use_sigmoid=True, loss_weight=1.0,
                                                   gamma=2.0, alpha=0.5, reduction='mean'),
                                     loss_bbox=dict(type='L1Loss', loss_weight=1.0))))


def _max_iou(pos, neg, min_pos, low_quality=None, nearest3d=False):
    d = dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=min_pos,
             ignore_iof_thr=-1)
    if low_quality is not None:
        d['match_low_quality'] = low_quality
    if nearest3d:
        d['iou_calculator'] = dict(type='BboxOverlapsNearest3D')
    return d


def frcnn_train_cfg():
    """split_0.py:440-478 (student.detector_2d)."""
    sampler = lambda num, frac, add_gt: dict(type='RandomSampler', num=num, pos_fraction=frac,
                                             neg_pos_ub=-1, add_gt_as_proposals=add_gt)
    return dict(
        rpn=dict(assigner=_max_iou(0.7, 0.3, 0.3, True), sampler=sampler(256, 0.5, False),
                 allowed_border=-1, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_pre=2000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7),
                          min_bbox_size=0),
        rcnn=dict(assigner=_max_iou(0.5, 0.5, 0.5, False), sampler=sampler(512, 0.25, True),
                  pos_weight=-1, debug=False))


def frcnn_test_cfg():
    """split_0.py:507-529"""
    return dict(rpn=dict(nms_pre=1000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7),
                         min_bbox_size=0),
                rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100))


def _hung_assigner():
    return dict(type='ModHungarianAssigner',
                cls_cost=dict(type='DoubleSidedFocalLossCost', weight=2.0),
                reg_cost=dict(type='BBoxL1Cost', weight=5.0),
                iou_cost=dict(type='IoUCost', iou_mode='giou', weight=2.0))


def _xf(kind, reverse, metas, src, dst):
    return dict(type='BboxesTransform_%s' % kind, reverse=reverse, img_metas=metas, in_bboxes_key=src,
                out_bboxes_key=dst)


def detmatch_ssl_cfg(class_names=CLASS_NAMES, with_vis=True):
    """ssl_cfg of split_0.py:200-437: the labeled chain (2 modules) and the unlabeled chain."""
    nms2d = lambda thr: dict(score_thr=thr, nms_pre=-1, max_num=100, iou_thr=0.5)
    T3, T2 = 'tea.3d_bboxes_nms', 'tea.2d_bboxes_nms'
    unl = [
        dict(type='Opd_SimpleTest_3D', ssl_obj_attr='teacher.detector_3d', batch_dict_key='tea',
             out_bboxes_key='3d_bboxes_nms'),
        _xf('3D', True, 'tea.img_metas', T3, T3 + '_no_aug'),
        _xf('3D', False, 'stu.img_metas', T3 + '_no_aug', T3 + '_stu_aug'),
        dict(type='MaxScoreFilter', cls_includes_bg_pred=False, score_thr=0.1,
             in_bboxes_key=T3 + '_no_aug', out_bboxes_key=T3 + '_no_aug_sc_filt'),
        dict(type='SimpleTest_2D', ssl_obj_attr='teacher.detector_2d', batch_dict_key='tea',
             out_bboxes_key='2d_bboxes'),
        dict(type='BboxesNMS_2D', nms_cfg=nms2d(0.05), cls_includes_bg_pred=True, batch_dict_key='tea',
             in_bboxes_key='2d_bboxes', out_bboxes_key='2d_bboxes_nms'),
        _xf('2D', True, 'tea.img_metas', T2, T2 + '_no_aug'),
        dict(type='MaxScoreFilter', cls_includes_bg_pred=True, score_thr=0.1,
             in_bboxes_key=T2 + '_no_aug', out_bboxes_key=T2 + '_no_aug_sc_filt'),
        dict(type='FusionHungarianMatching', assigner_cfg=_hung_assigner(), cost_thr=-1.5,
             img_metas='stu.img_metas', cls_includes_bg_pred_3d=False, cls_includes_bg_pred_2d=True,
             in_bboxes_3d_key=T3 + '_no_aug_sc_filt', in_bboxes_2d_key=T2 + '_no_aug_sc_filt',
             out_bboxes_3d_key=T3 + '_no_aug_hung', out_bboxes_2d_key=T2 + '_no_aug_hung'),
        _xf('3D', False, 'stu.img_metas', T3 + '_no_aug_hung', T3 + '_stu_aug_hung'),
        _xf('2D', False, 'stu.img_metas', T2 + '_no_aug_hung', T2 + '_stu_aug_hung'),
        dict(type='DetachBboxes', in_bboxes_key=T3 + '_stu_aug_hung', out_bboxes_key=T3 + '_stu_aug_hung_dtch'),
        dict(type='DetachBboxes', in_bboxes_key=T2 + '_stu_aug_hung', out_bboxes_key=T2 + '_stu_aug_hung_dtch'),
        dict(type='Opd_HardPseudoLabel_3D', score_thr=0.1, ssl_obj_attr='student.detector_3d',
             target_bboxes_key=T3 + '_stu_aug_hung_dtch', target_batch_dict_key='stu',
             out_bboxes_key='3d_bboxes_nms', no_nms=False),
        dict(type='HardPseudoLabel_2D', score_thr=0.1, cls_includes_bg_pred=True,
             loss_detach_keys=['loss_rpn_bbox', 'loss_bbox'], ssl_obj_attr='student.detector_2d',
             target_bboxes_key=T2 + '_stu_aug_hung_dtch', target_img_key='stu.img',
             target_img_metas_key='stu.img_metas', name='hard_pseudo_2d', weight=4),
        dict(type='Bboxes3DTo2D', img_metas='stu.img_metas', in_bboxes_key='stu.3d_bboxes_nms',
             out_bboxes_key='stu.3d_bboxes_nms_2d_proj'),
        dict(type='BboxesNMS_2D', nms_cfg=nms2d(0.1), cls_includes_bg_pred=False, batch_dict_key='stu',
             in_bboxes_key='3d_bboxes_nms_2d_proj', out_bboxes_key='3d_bboxes_nms_2d_proj_2d_nms'),
        dict(type='DetachBboxes', in_bboxes_key=T2 + '_no_aug_hung', out_bboxes_key=T2 + '_no_aug_hung_dtch'),
        dict(type='FusionHungarianMatching', assigner_cfg=_hung_assigner(), cost_thr=-1.5,
             img_metas='stu.img_metas', cls_includes_bg_pred_3d=False, cls_includes_bg_pred_2d=True,
             in_bboxes_3d_key='stu.3d_bboxes_nms_2d_proj_2d_nms',
             in_bboxes_2d_key=T2 + '_no_aug_hung_dtch',
             out_bboxes_3d_key='stu.3d_bboxes_nms_2d_proj_2d_nms_hung',
             out_bboxes_2d_key=T2 + '_no_aug_hung_dtch_hung', project_3d_to_2d=False),
        _xf('2D', False, 'stu.img_metas', 'stu.3d_bboxes_nms_2d_proj_2d_nms_hung',
            'stu.3d_bboxes_nms_2d_proj_2d_nms_hung_stu_aug'),
        _xf('2D', False, 'stu.img_metas', T2 + '_no_aug_hung_dtch_hung',
            T2 + '_no_aug_hung_dtch_hung_stu_aug'),
        dict(type='HungarianConsistency',
             loss_cls_cfg=dict(type='FocalLoss', reduction='mean', loss_weight=1.0),
             loss_iou_cfg=dict(type='GIoULoss', reduction='mean', loss_weight=1.0),
             loss_l1_cfg=dict(type='L1Loss', reduction='mean', loss_weight=1.0),
             loss_weights_cfg=dict(cls_loss=2, l1_loss=20, iou_loss=2),
             in_bboxes_key='stu.3d_bboxes_nms_2d_proj_2d_nms_hung_stu_aug',
             target_bboxes_key=T2 + '_no_aug_hung_dtch_hung_stu_aug',
             cls_includes_bg_pred_in=False, cls_includes_bg_pred_target=True,
             target_img_metas_key='stu.img_metas', name='2D_to_3D_hung'),
        dict(type='NumPreds', bboxes_key=T3 + '_stu_aug_hung_dtch', out_name='num_tea_hung'),
        dict(type='NumPreds', bboxes_key=T2 + '_no_aug_hung_dtch_hung', out_name='2D_to_3D_hung'),
    ]
    if with_vis:
        unl.append(dict(type='Vis3D', vis_idxs='data/kitti/ssl_splits/kitti_infos_train_unlab_0.01_0.pkl',
                        vis_idxs_interval=50, batch_dict_key='stu', stu_bboxes_key='stu.3d_bboxes_nms',
                        tea_bboxes_key=T3 + '_stu_aug_hung_dtch', out_name_prefix='tea',
                        class_names=class_names))
    lab = [dict(type='Opd_Supervised_3D', ssl_obj_attr='student.detector_3d', batch_dict_key='stu'),
           dict(type='TwoStageSupervised_2D', loss_detach_keys=[], ssl_obj_attr='student.detector_2d',
                batch_dict_key='stu')]
    return dict(labeled=lab, unlabeled=unl)


def confthr_pvrcnn_ssl_cfg():
    """ssl_cfg of configs/detmatch/001/confthr_pvrcnn/split_0.py: 3D-only confidence thresholding."""
    T3 = 'tea.3d_bboxes_nms'
    return dict(
        labeled=[dict(type='Opd_Supervised_3D', ssl_obj_attr='student.detector_3d', batch_dict_key='stu')],
        unlabeled=[
            dict(type='Opd_SimpleTest_3D', ssl_obj_attr='teacher.detector_3d', batch_dict_key='tea',
                 out_bboxes_key='3d_bboxes_nms'),
            _xf('3D', True, 'tea.img_metas', T3, T3 + '_no_aug'),
            _xf('3D', False, 'stu.img_metas', T3 + '_no_aug', T3 + '_stu_aug'),
            dict(type='Opd_HardPseudoLabel_3D', score_thr=0.3, ssl_obj_attr='student.detector_3d',
                 target_bboxes_key=T3 + '_stu_aug', target_batch_dict_key='stu'),
            dict(type='NumPreds', bboxes_key=T3 + '_stu_aug', out_name='tea')])


def confthr_frcnn_ssl_cfg(class_names=CLASS_NAMES, with_vis=True):
    """ssl_cfg of configs/detmatch/001/confthr_frcnn/split_0.py: 2D-only confidence thresholding
    (teacher Faster R-CNN boxes above 0.7 become hard pseudo labels of the student)."""
    T2 = 'tea.2d_bboxes_nms'
    unl = [
        dict(type='SimpleTest_2D', ssl_obj_attr='teacher.detector_2d', batch_dict_key='tea',
             out_bboxes_key='2d_bboxes'),
        dict(type='BboxesNMS_2D', nms_cfg=dict(nms_pre=-1, score_thr=0.7, max_num=100, iou_thr=0.5),
             cls_includes_bg_pred=True, batch_dict_key='tea', in_bboxes_key='2d_bboxes',
             out_bboxes_key='2d_bboxes_nms'),
        _xf('2D', True, 'tea.img_metas', T2, T2 + '_no_aug'),
        _xf('2D', False, 'stu.img_metas', T2 + '_no_aug', T2 + '_stu_aug'),
        dict(type='DetachBboxes', in_bboxes_key=T2 + '_stu_aug', out_bboxes_key=T2 + '_stu_aug_dtch'),
        dict(type='HardPseudoLabel_2D', score_thr=0.7, cls_includes_bg_pred=True,
             loss_detach_keys=['loss_rpn_bbox', 'loss_bbox'], ssl_obj_attr='student.detector_2d',
             target_bboxes_key=T2 + '_stu_aug_dtch', target_img_key='stu.img',
             target_img_metas_key='stu.img_metas', name='hard_pseudo_2d', weight=1),
        dict(type='MaxScoreFilter', cls_includes_bg_pred=True, score_thr=0.7,
             in_bboxes_key=T2 + '_stu_aug_dtch', out_bboxes_key=T2 + '_stu_aug_dtch_filt'),
        dict(type='NumPreds', bboxes_key=T2 + '_stu_aug_dtch_filt', out_name='num_tea'),
    ]
    if with_vis:
        unl.append(dict(type='Vis2D_Kitti', class_names=class_names, batch_dict_key='stu',
                        tea_bboxes_key=T2 + '_stu_aug_dtch_filt', stu_bboxes_key=None,
                        vis_idxs='data/kitti/ssl_splits/kitti_infos_train_unlab_0.01_0.pkl',
                        vis_idxs_interval=50, out_name_prefix='Vis_2D_Tea'))
    return dict(labeled=[dict(type='TwoStageSupervised_2D', loss_detach_keys=[],
                              ssl_obj_attr='student.detector_2d', batch_dict_key='stu')],
                unlabeled=unl)


def pretrain_pvrcnn_schedule(batch_size=8, max_epochs=40):
    """configs/detmatch/001/pretrain_pvrcnn/split_0.py:320-336 (supervised PV-RCNN pre-training)."""
    return dict(
        optimizer=dict(type='AdamW', lr=0.001 / 2 * batch_size, betas=(0.9, 0.99), weight_decay=0.01),
        optimizer_config=dict(grad_clip=dict(max_norm=10, norm_type=2)),
        lr_config=dict(policy='cyclic', target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4),
        momentum_config=dict(policy='cyclic', target_ratio=(0.85 / 0.95, 1), cyclic_times=1,
                             step_ratio_up=0.4),
        runner=dict(type='EpochBasedRunner', max_epochs=max_epochs))


def pretrain_frcnn_schedule(batch_size=8, max_epochs=12):
    """configs/detmatch/001/pretrain_frcnn/split_0.py:185-195 (supervised Faster R-CNN pre-training)."""
    return dict(
        optimizer=dict(type='SGD', lr=0.02 / 2 * batch_size, momentum=0.9, weight_decay=0.0001),
        optimizer_config=dict(grad_clip=None),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001,
                       step=[8, 10]),
        runner=dict(type='EpochBasedRunner', max_epochs=max_epochs))


def _pcdet_3d_train_cfg():
    """train_cfg.student.detector_3d of split_0.py:480-504 (mm3d-style; unused by OpenPCDetDetector)."""
    n3 = lambda pos, neg: _max_iou(pos, neg, neg, nearest3d=True)
    return dict(assigner=[n3(0.35, 0.2), n3(0.35, 0.2), n3(0.6, 0.45)], allowed_border=0, pos_weight=-1,
                debug=False)


WAYMO_POINT_CLOUD_RANGE = [-75.2, -75.2, -2, 75.2, 75.2, 4]
WAYMO_VOXEL_SIZE = [0.1, 0.1, 0.15]


def detmatch_kitti_model(ssl_cfg=None, pretrained=None, det3d_kwargs=None):
    """model = dict(type='SSL', ...) of configs/detmatch/001/detmatch/split_0.py:34-531.
    `det3d_kwargs` re-parameterises the 3D detector's geometry (BASELINE.json configs[4]: the
    Waymo-SHAPED synthetic run uses WAYMO_POINT_CLOUD_RANGE / WAYMO_VOXEL_SIZE / max_voxels 150000 —
    upstream OpenPCDet's Waymo PV-RCNN convention; the reference ships no such config)."""
    det3d = pvrcnn_kitti_model(**(det3d_kwargs or {}))
    test = dict(detector_2d=frcnn_test_cfg(), detector_3d=dict())
    import copy
    return dict(
        type='SSL', pretrained=pretrained,
        model_cfg=dict(type='MMDetector', detector_2d=frcnn_kitti_model(), detector_3d=det3d),
        ssl_cfg=ssl_cfg if ssl_cfg is not None else detmatch_ssl_cfg(),
        train_cfg=dict(
            ssl=dict(ema_params=dict(ema_decay=0.999, true_avg_rampup=True, rampup_start_decay=0.99),
                     weight_params=dict(weight=1), set_teacher_eval=True),
            teacher=None,
            student=dict(detector_2d=frcnn_train_cfg(), detector_3d=_pcdet_3d_train_cfg())),
        test_cfg=dict(teacher=copy.deepcopy(test), student=copy.deepcopy(test)))


def detmatch_schedule(batch_size=4, num_unlabeled_samples=1, max_iters=5000):
    """optimizer / optimizer_config / lr_config / runner / custom_hooks of split_0.py:827-868."""
    lr_3d = 0.001 / 2 * batch_size * (1 + num_unlabeled_samples) * 10
    lr_2d = 0.02 / 2 * batch_size * (1 + num_unlabeled_samples)
    return dict(
        optimizer={'constructor': 'HybridOptimizerConstructor',
                   'student.detector_3d': dict(type='AdamW', lr=lr_3d, betas=(0.95, 0.99),
                                               weight_decay=0.01, step_interval=1),
                   'student.detector_2d': dict(type='SGD', lr=lr_2d, momentum=0.9, weight_decay=0.0001,
                                               step_interval=1),
                   'teacher': dict(type='SGD', lr=1e-9, momentum=0.9, weight_decay=0.0001,
                                   step_interval=1)},
        optimizer_config=dict(grad_clip=dict(max_norm=10, norm_type=2)),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[]),
        runner=dict(type='IterBasedSSLRunner', max_iters=max_iters),
        custom_hooks=[dict(type='ModelIterEpochHook'), dict(type='WandbVisHook')])


# --------------------------------------------------------------------------------------------
# data section (configs/detmatch/001/detmatch/split_0.py:534-824), value for value
def _photometric_chain():
    erase = lambda p, scale, ratio: dict(type='TVRandomErasing', p=p, scale=scale, ratio=ratio, value='random')
    return [
        dict(type='TVToPILImage'),
        dict(type='RandomAppliedTrans',
             transforms=[dict(type='TVColorJitter', brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1)],
             p=0.8),
        dict(type='TVRandomGrayscale', p=0.2),
        dict(type='RandomAppliedTrans', transforms=[dict(type='GaussianBlur', sigma_min=0.1, sigma_max=2.0)],
             p=0.5),
        dict(type='TVToTensor'),
        erase(0.7, (0.05, 0.2), (0.3, 3.3)), erase(0.5, (0.02, 0.2), (0.1, 6)), erase(0.3, (0.02, 0.2), (0.05, 8)),
        dict(type='TVToPILImage'),
        dict(type='ToNumpy'),
    ]


def detmatch_pipelines(data_root='data/kitti/', db_info_path=None, class_names=CLASS_NAMES,
                       point_cloud_range=POINT_CLOUD_RANGE):
    """The seven pipeline lists of the DetMatch recipe, keyed as in the reference config."""
    file_client_args = dict(backend='disk')
    norm = dict(mean=[103.530, 116.280, 123.675], std=[1.0, 1.0, 1.0], to_rgb=False)
    db_sampler = dict(data_root=data_root, info_path=db_info_path, rate=1.0,
                      prepare=dict(filter_by_difficulty=[-1],
                                   filter_by_min_points=dict(Car=5, Pedestrian=5, Cyclist=5)),
                      classes=class_names, use_road_plane=True, limit_whole_scene=False,
                      sample_groups=dict(Car=15, Pedestrian=10, Cyclist=10))
    load_img = dict(type='LoadImageFromFile')
    load_pts = dict(type='LoadPointsFromFile', coord_type='LIDAR', load_dim=4, use_dim=4,
                    file_client_args=file_client_args)
    resize = dict(type='Resize', img_scale=[(640, 192), (2560, 768)], multiscale_mode='range', keep_ratio=True)
    flip = dict(type='RandomFlip3D', flip_ratio_bev_horizontal=0.5)
    grst = dict(type='GlobalRotScaleTrans', rot_range=[-0.78539816, 0.78539816], scale_ratio_range=[0.95, 1.05])
    prf = dict(type='PointsRangeFilter', point_cloud_range=point_cloud_range)
    tail = lambda keys: [dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32),
                         dict(type='DefaultFormatBundle3D', class_names=class_names),
                         dict(type='Collect3D', keys=keys)]
    teacher = [prf, dict(type='PointShuffle')] + tail(['points', 'img'])
    return dict(
        labeled_shared_pipeline=[
            load_img, load_pts,
            dict(type='LoadAnnotations3D', with_bbox_3d=True, with_label_3d=True, with_bbox=True,
                 with_label=True, file_client_args=file_client_args),
            dict(type='ObjectSample', db_sampler=db_sampler), resize, flip],
        labeled_student_pipeline=[grst, prf, dict(type='ObjectRangeFilter', point_cloud_range=point_cloud_range),
                                  dict(type='PointShuffle')] + _photometric_chain() +
        tail(['points', 'gt_bboxes_3d', 'gt_labels_3d', 'img', 'gt_bboxes', 'gt_labels']),
        labeled_teacher_pipeline=teacher,
        unlabeled_shared_pipeline=[load_img, load_pts, resize, flip],
        unlabeled_student_pipeline=[grst, prf, dict(type='PointShuffle')] + _photometric_chain() +
        tail(['points', 'img']),
        unlabeled_teacher_pipeline=teacher,
        test_pipeline=[
            load_img, load_pts,
            dict(type='MultiScaleFlipAug3D', img_scale=(1280, 384), pts_scale_ratio=1, flip=False,
                 transforms=[dict(type='GlobalRotScaleTrans', rot_range=[0, 0], scale_ratio_range=[1., 1.],
                                  translation_std=[0, 0, 0]),
                             dict(type='Resize', keep_ratio=True), dict(type='RandomFlip3D'), prf,
                             dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32),
                             dict(type='DefaultFormatBundle3D', class_names=class_names, with_label=False),
                             dict(type='Collect3D', keys=['points', 'img'])])])


def detmatch_data(data_root='data/kitti/', split_folder='ssl_splits', split_frac=0.01, split_num=0, batch_size=4,
                  class_names=CLASS_NAMES, point_cloud_range=POINT_CLOUD_RANGE, lab_info=None, unlab_info=None,
                  db_info=None, val_info=None):
    """`data = dict(...)` of split_0.py:764-824."""
    fmt = lambda stem: data_root + '%s/%s_%s_%s.pkl' % (split_folder, stem, split_frac, split_num)
    lab_info = lab_info or fmt('kitti_infos_train_proj_3d_lab')
    unlab_info = unlab_info or fmt('kitti_infos_train_unlab')
    db_info = db_info or fmt('kitti_dbinfos_train_lab')
    val_info = val_info or data_root + 'kitti_infos_val.pkl'
    pipes = detmatch_pipelines(data_root, db_info, class_names, point_cloud_range)
    modality = dict(use_lidar=True, use_camera=True)
    kitti = lambda ann, pipe, **kw: dict(type='KittiDataset', data_root=data_root, ann_file=ann, split='training',
                                         pts_prefix='velodyne_reduced', pipeline=pipe, modality=modality,
                                         classes=class_names, box_type_3d='LiDAR', **kw)
    test = kitti(val_info, pipes['test_pipeline'], test_mode=True)
    return dict(
        samples_per_gpu=batch_size, workers_per_gpu=1,
        train_lab=dict(type='TS_SSL_Dataset',
                       dataset=dict(type='RepeatDataset', times=100,
                                    dataset=kitti(lab_info, pipes['labeled_shared_pipeline'], test_mode=False,
                                                  completely_remove_other_classes=True)),
                       student_pipeline=pipes['labeled_student_pipeline'],
                       teacher_pipeline=pipes['labeled_teacher_pipeline']),
        train_unlab=dict(type='TS_SSL_Dataset',
                         dataset=kitti(unlab_info, pipes['unlabeled_shared_pipeline'], test_mode=False,
                                       filter_empty_gt=False, completely_remove_other_classes=True),
                         student_pipeline=pipes['unlabeled_student_pipeline'],
                         teacher_pipeline=pipes['unlabeled_teacher_pipeline']),
        val=test, test=dict(test))
