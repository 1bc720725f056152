"""PVRCNN detector — pcdet/models/detectors/{detector3d_template,pv_rcnn}.py and
pcdet/models/__init__.py:16 (build_network).

Module names (vfe, backbone_3d, map_to_bev_module, pfe, backbone_2d, dense_head,
point_head, roi_head) and the `global_step` buffer follow the reference so that state
dicts are interchangeable.
"""

import os

import torch
import torch.nn as nn

from .. import iou3d_nms
from .backbones_2d import BaseBEVBackbone
from .backbones_3d import HeightCompression, MeanVFE, VoxelBackBone8x
from .config import ConfigDict
from .dense_heads import AnchorHeadSingle, PointHeadSimple, valid_gt_mask
from .pfe import VoxelSetAbstraction
from .roi_heads import PVRCNNHead, class_agnostic_nms_fixed_batch

_MODULES = {
    'MeanVFE': MeanVFE, 'VoxelBackBone8x': VoxelBackBone8x, 'HeightCompression': HeightCompression,
    'BaseBEVBackbone': BaseBEVBackbone, 'VoxelSetAbstraction': VoxelSetAbstraction,
    'AnchorHeadSingle': AnchorHeadSingle, 'PointHeadSimple': PointHeadSimple,
    'PVRCNNHead': PVRCNNHead,
}


# the key-point encoder (pfe) beside the BEV backbone on the side stream (PVRCNN.run_modules; module switch of the
# equality test)
# (training passes only: the teacher's encoder on the side stream as well measured 61.6 / 62.4 against 55.7 / 54.0 ms — it
# queues in front of the student's work there)
PFE_SIDE = [True]
_PFE_OUT = ('point_features', 'point_features_before_fusion', 'point_coords', 'point_batch_cnt')


def _on_device(batch_dict):
    p = batch_dict.get('points')
    return torch.is_tensor(p) and p.is_cuda


def _record_tree(value, stream):
    """record_stream for every device tensor of a (nested) value that another stream is about to read"""
    if torch.is_tensor(value):
        if value.is_cuda:
            value.record_stream(stream)
    elif hasattr(value, 'features') and hasattr(value, 'indices'):      # SparseConvTensor
        _record_tree(value.features, stream)
        _record_tree(value.indices, stream)
    elif isinstance(value, dict):
        for v in value.values():
            _record_tree(v, stream)
    elif isinstance(value, (list, tuple)):
        for v in value:
            _record_tree(v, stream)


class PVRCNN(nn.Module):
    """forward(batch_dict): vfe -> backbone_3d -> map_to_bev -> pfe -> backbone_2d ->
    dense_head -> point_head -> roi_head (detector3d_template.py:22-25, pv_rcnn.py:9-22)."""

    module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d',
                       'dense_head', 'point_head', 'roi_head']

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg if isinstance(model_cfg, ConfigDict) else ConfigDict(model_cfg)
        self.num_class = num_class
        self.dataset = dataset if isinstance(dataset, ConfigDict) else ConfigDict(dataset)
        self.class_names = self.dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.record_recall = False
        self.module_list = self.build_networks()

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        ds = self.dataset
        cfg = self.model_cfg
        info = dict(num_rawpoint_features=ds.point_feature_encoder.num_point_features,
                    num_point_features=ds.point_feature_encoder.num_point_features,
                    grid_size=ds.grid_size, point_cloud_range=ds.point_cloud_range,
                    voxel_size=ds.voxel_size)
        mods = []
        self.vfe = _MODULES[cfg.VFE.NAME](model_cfg=cfg.VFE,
                                          num_point_features=info['num_rawpoint_features'])
        info['num_point_features'] = self.vfe.get_output_feature_dim()
        mods.append(self.vfe)
        self.backbone_3d = _MODULES[cfg.BACKBONE_3D.NAME](
            model_cfg=cfg.BACKBONE_3D, input_channels=info['num_point_features'],
            grid_size=info['grid_size'])
        info['num_point_features'] = self.backbone_3d.num_point_features
        mods.append(self.backbone_3d)
        self.map_to_bev_module = _MODULES[cfg.MAP_TO_BEV.NAME](model_cfg=cfg.MAP_TO_BEV)
        info['num_bev_features'] = self.map_to_bev_module.num_bev_features
        mods.append(self.map_to_bev_module)
        self.pfe = _MODULES[cfg.PFE.NAME](
            model_cfg=cfg.PFE, voxel_size=info['voxel_size'],
            point_cloud_range=info['point_cloud_range'],
            num_bev_features=info['num_bev_features'],
            num_rawpoint_features=info['num_rawpoint_features'])
        info['num_point_features'] = self.pfe.num_point_features
        info['num_point_features_before_fusion'] = self.pfe.num_point_features_before_fusion
        mods.append(self.pfe)
        self.backbone_2d = _MODULES[cfg.BACKBONE_2D.NAME](model_cfg=cfg.BACKBONE_2D,
                                                          input_channels=info['num_bev_features'])
        info['num_bev_features'] = self.backbone_2d.num_bev_features
        mods.append(self.backbone_2d)
        self.dense_head = _MODULES[cfg.DENSE_HEAD.NAME](
            model_cfg=cfg.DENSE_HEAD, input_channels=info['num_bev_features'],
            num_class=self.num_class if not cfg.DENSE_HEAD.CLASS_AGNOSTIC else 1,
            class_names=self.class_names, grid_size=info['grid_size'],
            point_cloud_range=info['point_cloud_range'],
            predict_boxes_when_training=cfg.get('ROI_HEAD', False))
        mods.append(self.dense_head)
        use_before = cfg.POINT_HEAD.get('USE_POINT_FEATURES_BEFORE_FUSION', False)
        self.point_head = _MODULES[cfg.POINT_HEAD.NAME](
            model_cfg=cfg.POINT_HEAD,
            input_channels=info['num_point_features_before_fusion'] if use_before
            else info['num_point_features'],
            num_class=self.num_class if not cfg.POINT_HEAD.CLASS_AGNOSTIC else 1)
        mods.append(self.point_head)
        self.roi_head = _MODULES[cfg.ROI_HEAD.NAME](
            model_cfg=cfg.ROI_HEAD, input_channels=info['num_point_features'],
            num_class=self.num_class if not cfg.ROI_HEAD.CLASS_AGNOSTIC else 1)
        mods.append(self.roi_head)
        # the RoI head's proposal layer detaches the anchor head's boxes (roi_head_template.py:96-99) and
        # the final predictions are the RoI head's: nothing differentiates through the decoded anchors
        self.dense_head.boxes_detached_downstream = True
        # the anchor head's 1x1 convolutions are the tail of the BEV backbone's shape-static chain (graphs.py);
        # a plain dict entry, not a sub-module: the state-dict keys stay the reference's
        if hasattr(self.dense_head, 'conv_heads') and getattr(self, 'backbone_2d', None) is not None:
            self.backbone_2d.__dict__['fused_head'] = self.dense_head
        return mods

    def _order(self, batch_dict):
        """The reference's module order, except that the BEV backbone is issued BEFORE the key-point
        encoder while the key points are still being sampled on a side stream (`keypoints_async`): the
        backbone reads only `spatial_features`, which the encoder does not modify, so the order does not
        change any value — it gives the FPS kernel (6.6 ms per pass) 8 ms of convolutions to hide behind
        instead of stalling the main stream right after the sparse backbone."""
        mods = self.module_list
        if 'keypoints_async' in batch_dict:
            pfe, bev = getattr(self, 'pfe', None), getattr(self, 'backbone_2d', None)
            if pfe is not None and bev is not None and mods.index(bev) == mods.index(pfe) + 1:
                i = mods.index(pfe)
                mods = mods[:i] + [bev, pfe] + mods[i + 2:]
        return mods

    def run_modules(self, batch_dict, until=None):
        """The module chain.
        `until`: stop in front of this module; the rest of the chain is remembered in the batch dict and runs when
        run_modules is called on it again (SSL issues the label-independent part of a pass — everything in front
        of the first module that reads `gt_boxes` — before the pseudo-labels exist)."""
        todo = batch_dict.pop('_pending_modules', None)
        if todo is None:
            todo = list(self._order(batch_dict))
        pfe, bev, head = getattr(self, 'pfe', None), getattr(self, 'backbone_2d', None), getattr(self, 'dense_head', None)
        while todo:
            cur_module = todo[0]
            if until is not None and cur_module is until:
                batch_dict['_pending_modules'] = todo
                return batch_dict
            todo.pop(0)
            if PFE_SIDE[0] and self.training and cur_module is bev and todo and todo[0] is pfe and _on_device(batch_dict):
                # The key-point encoder reads the sparse features, the raw points and the BEV map that goes INTO the BEV
                # backbone — nothing the backbone produces — and only the point / RoI heads read what it writes
                # (pv_rcnn.py:9-22 runs them one after the other because a module list has no other order).  So the
                # encoder runs on the side stream beside the backbone's convolutions, and — autograd runs a node's
                # backward on the stream of its forward — its backward beside the backbone's.  Scheduling only.
                todo.pop(0)
                batch_dict = self._bev_and_pfe(bev, pfe, batch_dict)
                continue
            ev = batch_dict.get('_pfe_done') if cur_module is not head else None
            if ev is not None:                       # first reader of the encoder's outputs on this stream
                main = torch.cuda.current_stream()
                main.wait_event(ev)
                for k in _PFE_OUT:
                    if torch.is_tensor(batch_dict.get(k)):
                        batch_dict[k].record_stream(main)
                del batch_dict['_pfe_done']
            batch_dict = cur_module(batch_dict)
        return batch_dict

    def _bev_and_pfe(self, bev, pfe, batch_dict):
        from .. import _lib
        dev = batch_dict['points'].device
        main = torch.cuda.current_stream(dev)
        side = _lib.aux_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)
        feats = {}
        for k in ('spatial_features', 'points', 'voxel_coords'):
            if torch.is_tensor(batch_dict.get(k)):
                feats[k] = batch_dict[k]
        batch_dict = bev(batch_dict)                 # one chained call: the main lane has its work first
        side.wait_event(ready)
        with torch.cuda.stream(side):
            _record_tree([feats, batch_dict.get('multi_scale_3d_features')], side)
            batch_dict = pfe(batch_dict)
            done = torch.cuda.Event()
            done.record(side)
        batch_dict['_pfe_done'] = done
        return batch_dict

    def label_independent_until(self):
        """The first module of the chain that reads the ground truth in training mode (target assignment)."""
        return getattr(self, 'dense_head', None)

    def forward(self, batch_dict):
        batch_dict = self.run_modules(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def forward_issue(self, batch_dict):
        """Evaluation mode, everything up to the one read-back of post_processing; `forward_finish` completes it.
        forward(batch) == forward_finish(forward_issue(batch))."""
        assert not self.training
        return self.post_processing_issue(self.run_modules(batch_dict))

    def forward_finish(self, state):
        return self.post_processing_finish(state)

    def get_training_loss(self):
        """pv_rcnn.py:24-32"""
        loss_rpn, tb_dict = self.dense_head.get_loss()
        loss_point, tb_dict = self.point_head.get_loss(tb_dict)
        loss_rcnn, tb_dict = self.roi_head.get_loss(tb_dict)
        return loss_rpn + loss_point + loss_rcnn, tb_dict, {}

    def post_processing(self, batch_dict, no_nms=False):
        """detector3d_template.py:176-309 (MULTI_CLASSES_NMS False branch)."""
        return self.post_processing_finish(self.post_processing_issue(batch_dict, no_nms))

    def post_processing_issue(self, batch_dict, no_nms=False):
        """Pass 1 of post_processing: everything that stays on the device; the survivor counts start their way to
        the host (pinned buffer + event) without blocking."""
        cfg = self.model_cfg.POST_PROCESSING
        assert not cfg.NMS_CONFIG.MULTI_CLASSES_NMS
        batch_size = batch_dict['batch_size']
        recall_dict = {}
        pred_dicts = []
        staged = []
        # pass 1: everything that stays on the device (scores, labels, fixed-size NMS), over the batch
        all_cls = batch_dict['batch_cls_preds']
        assert all_cls.dim() == 3 and all_cls.shape[2] in [1, self.num_class]
        if not batch_dict['cls_preds_normalized']:
            all_cls = torch.sigmoid(all_cls)
        all_cls, all_labels = torch.max(all_cls, dim=-1)
        all_sel = all_valid = None
        if not no_nms:
            all_sel, all_valid = class_agnostic_nms_fixed_batch(
                all_cls.detach(), batch_dict['batch_box_preds'].detach(), cfg.NMS_CONFIG,
                score_thresh=cfg.SCORE_THRESH)
        for index in range(batch_size):
            box_preds = batch_dict['batch_box_preds'][index]
            src_box_preds = box_preds
            src_cls_preds = batch_dict['batch_cls_preds'][index]
            cls_preds, label_preds = all_cls[index], all_labels[index]
            if batch_dict.get('has_class_labels', False):
                label_key = 'roi_labels' if 'roi_labels' in batch_dict else 'batch_pred_labels'
                label_preds = batch_dict[label_key][index]
                sem_scores = batch_dict['roi_scores'][index]
                sem_scores_full = batch_dict['roi_scores_full'][index]
            else:
                label_preds = label_preds + 1
                sem_scores = cls_preds
                sem_scores_full = src_cls_preds
            sel = valid = None
            if not no_nms:
                sel, valid = all_sel[index], all_valid[index]
            staged.append((box_preds, src_box_preds, cls_preds, src_cls_preds, label_preds, sem_scores,
                           sem_scores_full, sel, valid))
        # (the survivor counts stay on the device until post_processing_finish reads them back)
        counts = all_valid.sum(dim=1) if not no_nms else None
        return dict(staged=staged, counts=counts, no_nms=no_nms, batch_dict=batch_dict, recall_dict=recall_dict,
                    pred_dicts=pred_dicts)

    def post_processing_finish(self, state):
        """Pass 2: the ONE read-back of the call — how many boxes survive per sample (the reference returns
        variable-length tensors too, model_nms_utils.py:20) — and the variable-length records."""
        cfg = self.model_cfg.POST_PROCESSING
        staged, no_nms, batch_dict = state['staged'], state['no_nms'], state['batch_dict']
        recall_dict, pred_dicts = state['recall_dict'], state['pred_dicts']
        if not no_nms:
            keep_counts = state['counts'].tolist()
        # pass 2: variable-length records
        for index, (box_preds, src_box_preds, cls_preds, src_cls_preds, label_preds, sem_scores,
                    sem_scores_full, sel, valid) in enumerate(staged):
            if no_nms:
                selected = torch.arange(len(cls_preds), device=cls_preds.device)
                selected_scores = cls_preds
            else:
                selected = sel[:int(keep_counts[index])]
                selected_scores = cls_preds.index_select(0, selected)
            if cfg.OUTPUT_RAW_SCORE:
                selected_scores = torch.max(src_cls_preds, dim=-1)[0].index_select(0, selected)
            # index_select: same values as t[selected]; its backward is zeros + index_add instead of the
            # sort-based index_put of advanced indexing (these carry the consistency losses' gradients)
            record = {'pred_boxes': box_preds.index_select(0, selected), 'pred_scores': selected_scores,
                      'pred_labels': label_preds.index_select(0, selected),
                      'pred_sem_scores': torch.sigmoid(sem_scores.index_select(0, selected)),
                      'pred_sem_scores_full': torch.sigmoid(sem_scores_full.index_select(0, selected))}
            pred_dicts.append(record)
            if self.record_recall:
                recall_dict = self.generate_recall_record(
                    record['pred_boxes'] if 'rois' not in batch_dict else src_box_preds,
                    recall_dict, index, batch_dict, cfg.RECALL_THRESH_LIST)
        return pred_dicts, recall_dict

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None,
                               thresh_list=None):
        """detector3d_template.py:312-354; counters are 0-d device tensors (no .item())."""
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {'gt': 0}
            for t in thresh_list:
                recall_dict['roi_%s' % str(t)] = 0
                recall_dict['rcnn_%s' % str(t)] = 0
        valid = valid_gt_mask(gt_boxes[None])[0]
        if box_preds.shape[0] > 0:
            iou_rcnn = iou3d_nms.boxes_iou3d_gpu(box_preds[:, 0:7], gt_boxes[:, 0:7])
        else:
            iou_rcnn = None
        iou_roi = iou3d_nms.boxes_iou3d_gpu(rois[:, 0:7], gt_boxes[:, 0:7]) if rois is not None \
            else None
        for t in thresh_list:
            if iou_rcnn is not None:
                recall_dict['rcnn_%s' % str(t)] = recall_dict['rcnn_%s' % str(t)] + \
                    ((iou_rcnn.max(dim=0)[0] > t) & valid).sum()
            if iou_roi is not None:
                recall_dict['roi_%s' % str(t)] = recall_dict['roi_%s' % str(t)] + \
                    ((iou_roi.max(dim=0)[0] > t) & valid).sum()
        recall_dict['gt'] = recall_dict['gt'] + valid.sum()
        return recall_dict


_DETECTORS = {'PVRCNN': PVRCNN}


def build_network(model_cfg, num_class, dataset):
    """pcdet/models/__init__.py:16-20 / detectors/__init__.py build_detector."""
    model_cfg = model_cfg if isinstance(model_cfg, ConfigDict) else ConfigDict(model_cfg)
    if model_cfg.NAME not in _DETECTORS:
        raise NotImplementedError('%s: only PVRCNN is on the DetMatch hot path' % model_cfg.NAME)
    return _DETECTORS[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
