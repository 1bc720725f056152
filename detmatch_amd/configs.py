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
