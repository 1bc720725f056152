"""SSL modules — the processors and consumers that configs/detmatch/*/detmatch/*.py chain
inside SSL.forward_train (mmdet3d/models/ssl_modules/{processors,consumers}/*.py).

Every module is `forward(ssl_obj, batch_dict) -> batch_dict`; dotted keys address nested
dicts ("tea.3d_simple_test").  Box lists are list[tuple(boxes, scores, *extras)], one tuple
per sample.  Names, constructor arguments and key semantics follow the reference so its config
files load unchanged (tests/test_ssl_config.py).
"""
import numpy as np
import torch

from .bbox_utils import (apply_3d_transformation_bboxes, bbox_2d_transform, bbox_3d_to_bbox_2d,
                         filter_by_nms_2d, filter_by_nms_2d_masked, mlvl_get, mlvl_getattr, mlvl_set, take, unaug_project_boxes)
from ..devconst import const
from ..fused import on as fused_on
from .box3d import LiDARInstance3DBoxes
from .losses import FocalLoss, MSELoss, bbox_xyxy_to_cxcywh
from .openpcdet import pcdet_to_mm3d_boxes
from .registry import SSL_MODULES, build_assigner, build_loss
from .ssl import add_prefix


# ------------------------------------------------------------------ helpers
def _split(entry):
    """entry is a tuple (boxes, *rest) or a bare box tensor/object."""
    if isinstance(entry, tuple):
        return entry[0], entry[1:], True
    return entry, (), False


def _join(boxes, rest, was_tuple):
    return (boxes,) + tuple(rest) if was_tuple else boxes


class Masked(object):
    """One sample's box-list entry — (boxes, scores, *extras) or bare boxes — whose rows count only where `keep` (a device
    bool per row) says so: the output of a FILTER that has not been compacted.  Compaction (`nonzero` -> data-dependent
    shape) is what makes the host wait for the device in the pseudo-label glue (the reference pays it per module and per
    sample: processors_fusion.py:29-46, bbox_utils.py:286-347, processors_3d.py:140-150); the modules between a filter
    and the Hungarian matching only map rows (augmentation replay) or filter again, and the matching reads its cost
    matrix back to the host anyway — so the masks ride along and are resolved THERE, in the same read-back.
    `entry()` compacts (one synchronising nonzero), for consumers that need true lengths."""
    __slots__ = ('full', 'keep', '_entry')

    def __init__(self, full, keep):
        self.full, self.keep, self._entry = full, keep, None

    def entry(self):
        if self._entry is None:
            ki = self.keep.nonzero(as_tuple=False).squeeze(1)
            boxes, rest, tup = _split(self.full)
            self._entry = _join(take(boxes, ki), [take(t, ki) for t in rest], tup)
        return self._entry


def plain(entries):
    """A box list with every lazily filtered entry compacted (what the reference's modules hand each other)."""
    if entries is None:
        return None
    return [e.entry() if isinstance(e, Masked) else e for e in entries]


def _lazy(t):
    """Filters stay lazy on the MI355X only (host tensors: compact at once, as the reference does)."""
    return fused_on() and isinstance(t, torch.Tensor) and t.is_cuda


def _lap_host(cost_host):
    """dm_lap_host on a host (n, m) float32 matrix -> (rows, cols) int64, rows ascending."""
    from .losses import linear_sum_assignment_host
    return linear_sum_assignment_host(cost_host)


def _fg_scores(scores, includes_bg):
    return scores[:, :-1] if includes_bg else scores


def _accumulate(ssl_obj, batch_dict, losses, prefer_sup):
    tgt = 'sup_losses' if (prefer_sup and 'sup_losses' in batch_dict) else 'ssl_losses'
    batch_dict[tgt] = ssl_obj._sum_update_losses(batch_dict[tgt], losses)
    return batch_dict


def _pred_dicts_to_tuples(pred_dicts):
    """pcdet predictions -> list[(LiDARInstance3DBoxes, per-class sigmoid scores)]
    (consumers/openpcdet.py:74-93)."""
    res = []
    for pd in pred_dicts:
        boxes = pcdet_to_mm3d_boxes(pd['pred_boxes'])
        assert len(boxes.tensor) == len(pd['pred_sem_scores_full'])
        res.append((boxes, pd['pred_sem_scores_full']))
    return res


def _threshold_pseudo(entries, score_thr, includes_bg, empty_boxes, dense=False):
    """Hard pseudo labels: boxes whose max foreground score exceeds score_thr, label = argmax
    (consumers/openpcdet.py:139-158, consumers_2d.py:84-103).  Boolean indexing is the one
    place the teacher path produces data-dependent shapes.
    dense=True (a consumer that takes padded ground truth: OpenPCDetDetector.add_gt turns rows whose label is out of
    range into padding and moves them behind the valid rows, in order): boxes are NOT selected, rejected rows get the
    label -1 — the same ground truth without the two synchronising selections per sample."""
    labels, boxes = [], []
    for cur_boxes, cur_scores in plain(entries):
        if dense and len(cur_scores) and _lazy(cur_scores):
            top, lab = _fg_scores(cur_scores, includes_bg).max(dim=1)
            labels.append(torch.where(top > score_thr, lab, torch.full_like(lab, -1)))
            boxes.append(cur_boxes)
            continue
        if len(cur_scores) == 0:
            labels.append(cur_scores.new_zeros((0,), dtype=torch.long))
            boxes.append(empty_boxes(cur_scores))
            continue
        top, lab = _fg_scores(cur_scores, includes_bg).max(dim=1)
        keep = top > score_thr
        labels.append(lab[keep])
        boxes.append(cur_boxes[keep])
    return boxes, labels


class _Early(object):
    """Holder of a pass that was issued early (kept opaque: a stream-lane batch dict wraps plain dicts)."""
    __slots__ = ('value',)

    def __init__(self, value):
        self.value = value


# ------------------------------------------------------------------ 3D (OpenPCDet) modules
@SSL_MODULES.register_module()
class Opd_SimpleTest_3D(object):
    """consumers/openpcdet.py:15-95: run the detector in its CURRENT mode on [key]['points']
    and store un-thresholded (boxes, sigmoid class scores)."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    # scheduling hints for SSL.forward_train (lane mode 'glue'): reads only the raw batch -> may be
    # issued ahead of the labeled chain; ends with a host read-back (post_processing)
    hoistable = True
    has_readback = True

    def __init__(self, ssl_obj_attr='teacher', batch_dict_key='tea',
                 out_bboxes_key='3d_simple_test'):
        self.ssl_obj_attr = ssl_obj_attr
        self.batch_dict_key = batch_dict_key
        self.out_bboxes_key = out_bboxes_key

    def prefetch(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if hasattr(detector, 'prepare_geometry'):
            detector.prepare_geometry(cur['points'], cur['img_metas'])

    def prefetch_steps(self, ssl_obj, batch_dict, ws_tag):
        """prefetch as a generator (OpenPCDetDetector.prepare_geometry_steps), or None."""
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if hasattr(detector, 'prepare_geometry_steps'):
            return detector.prepare_geometry_steps(cur['points'], cur['img_metas'], ws_tag)
        return None

    def issue_early(self, ssl_obj, batch_dict):
        """Scheduling only: the whole pass up to its read-back, issued before the host needs the result
        (SSL.forward_train calls this for the modules of the unlabeled chain right after the labeled chain), so
        that the read-back in forward() finds the device done instead of stalling the issue of everything else."""
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        model = getattr(detector, 'model', None)
        if model is None or model.training or not hasattr(model, 'forward_issue'):
            return
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if '_early.' + self.out_bboxes_key in cur:       # issued already (SSL: the teacher's passes ahead of everything else)
            return
        batch = detector._base_batch(cur['points'], cur['img_metas'])
        cur['_early.' + self.out_bboxes_key] = _Early(model.forward_issue(batch))

    def forward(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        state = cur.pop('_early.' + self.out_bboxes_key, None) if isinstance(cur, dict) else None
        if state is not None:
            pred_dicts, _ = detector.model.forward_finish(state.value)
        else:
            batch = detector._base_batch(cur['points'], cur['img_metas'])
            pred_dicts, _ = detector.model(batch)
        cur[self.out_bboxes_key] = _pred_dicts_to_tuples(pred_dicts)
        return batch_dict


@SSL_MODULES.register_module()
class Opd_HardPseudoLabel_3D(object):
    """consumers/openpcdet.py:97-213: threshold teacher boxes into pseudo GT, run the student's
    training forward on them, and optionally keep the student's (no-NMS) boxes."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    trunk_may_change_lane = True       # issue_early leaves an event behind its trunk: forward_issue may run on another stream

    def __init__(self, score_thr, cls_includes_bg_pred=False, loss_detach_keys=[],
                 ssl_obj_attr='student', target_bboxes_key='tea.placeholder',
                 target_batch_dict_key='stu', name='hard_pseudo_3d', weight=1,
                 out_bboxes_key=None, no_nms=True, box_dim=7):
        assert len(loss_detach_keys) == 0, 'Not supported yet, requires changes elsewhere.'
        self.score_thr = score_thr
        self.cls_includes_bg_pred = cls_includes_bg_pred
        self.loss_detach_keys = loss_detach_keys
        self.ssl_obj_attr = ssl_obj_attr
        self.target_bboxes_key = target_bboxes_key
        self.target_batch_dict_key = target_batch_dict_key
        self.name = name
        self.weight = weight
        self.out_bboxes_key = out_bboxes_key
        self.no_nms = no_nms
        self.box_dim = box_dim

    def prefetch(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.target_batch_dict_key)
        if hasattr(detector, 'prepare_geometry'):
            detector.prepare_geometry(cur['points'], cur['img_metas'])

    def prefetch_steps(self, ssl_obj, batch_dict, ws_tag):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.target_batch_dict_key)
        if hasattr(detector, 'prepare_geometry_steps'):
            return detector.prepare_geometry_steps(cur['points'], cur['img_metas'], ws_tag)
        return None

    def issue_early(self, ssl_obj, batch_dict):
        """Scheduling only: the label-independent part of the student's pass (voxel features, sparse backbone, BEV
        backbone, key-point encoder — everything in front of the first target assignment) issued before the
        pseudo-labels exist; forward() adds the labels and runs the rest.  Same kernels in the same order on the
        same stream as the undivided pass."""
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        model = getattr(detector, 'model', None)
        if model is None or not model.training or not hasattr(model, 'label_independent_until') \
                or model.label_independent_until() is None:
            return
        cur = mlvl_get(batch_dict, self.target_batch_dict_key)
        batch = detector._base_batch(cur['points'], cur['img_metas'])
        batch = model.run_modules(batch, until=model.label_independent_until())
        where = None
        if cur['points'][0].is_cuda:               # forward_issue may run on another stream (SSL: trunk on the 2D lane)
            stream = torch.cuda.current_stream(cur['points'][0].device)
            ev = torch.cuda.Event()
            ev.record(stream)
            where = (stream, ev)
        cur['_early.trunk3d'] = _Early((batch, where))

    def forward_issue(self, ssl_obj, batch_dict):
        """forward() up to the read-back of the student's own boxes (`out_bboxes_key`): pseudo labels, the rest of the
        pass, the losses, and the device half of post_processing.  SSL.forward_train issues the student's last 2D module
        between this and forward_finish when the two are independent — the host then does not sit in the read-back while
        the 2D lane has nothing queued."""
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.target_batch_dict_key)
        boxes, labels = _threshold_pseudo(
            mlvl_get(batch_dict, self.target_bboxes_key), self.score_thr,
            self.cls_includes_bg_pred,
            lambda s: LiDARInstance3DBoxes(s.new_zeros((0, self.box_dim))),
            dense=hasattr(detector, 'add_gt'))
        early = cur.pop('_early.trunk3d', None) if isinstance(cur, dict) else None
        if early is not None:
            trunk, where = early.value
            if where is not None:
                here = torch.cuda.current_stream(cur['points'][0].device)
                if here != where[0]:               # issued on another lane: order it, keep the allocator informed
                    from ..pcdet.detector import _record_tree
                    here.wait_event(where[1])
                    _record_tree({k: v for k, v in trunk.items() if k != '_pending_modules'}, here)
            batch = detector.add_gt(trunk, cur['points'], boxes, labels)
        else:
            batch = detector.train_to_openpcdet(cur['points'], cur['img_metas'], boxes, labels)
        batch = detector.model.run_modules(batch)
        loss, _, _ = detector.model.get_training_loss()
        batch_dict = _accumulate(ssl_obj, batch_dict, add_prefix(dict(loss=loss.mean()), self.name),
                                 prefer_sup=False)
        if self.out_bboxes_key is not None:
            if hasattr(detector.model, 'post_processing_issue'):
                cur['_pending.' + self.out_bboxes_key] = _Early(('state', detector.model.post_processing_issue(
                    batch, no_nms=self.no_nms)))
            else:
                cur['_pending.' + self.out_bboxes_key] = _Early(('done', detector.model.post_processing(
                    batch, no_nms=self.no_nms)))
        return batch_dict

    def forward_finish(self, ssl_obj, batch_dict):
        if self.out_bboxes_key is None:
            return batch_dict
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.target_batch_dict_key)
        kind, value = cur.pop('_pending.' + self.out_bboxes_key).value
        pred_dicts, _ = detector.model.post_processing_finish(value) if kind == 'state' else value
        cur[self.out_bboxes_key] = _pred_dicts_to_tuples(pred_dicts)
        return batch_dict

    def forward(self, ssl_obj, batch_dict):
        return self.forward_finish(ssl_obj, self.forward_issue(ssl_obj, batch_dict))


@SSL_MODULES.register_module()
class Opd_Supervised_3D(object):
    """consumers/openpcdet.py:215-253"""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    # forward() adds losses and nothing else: no later module reads tensors of this pass's autograd
    # graph, so SSL.forward_train may back-propagate these losses as soon as they exist
    self_contained_losses = True

    def __init__(self, ssl_obj_attr='student', batch_dict_key='stu', name='sup_3d', weight=1):
        self.ssl_obj_attr = ssl_obj_attr
        self.batch_dict_key = batch_dict_key
        self.name = name
        self.weight = weight

    def prefetch(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if hasattr(detector, 'prepare_geometry'):
            detector.prepare_geometry(cur['points'], cur['img_metas'])

    def prefetch_steps(self, ssl_obj, batch_dict, ws_tag):
        """prefetch as a generator (OpenPCDetDetector.prepare_geometry_steps), or None."""
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if hasattr(detector, 'prepare_geometry_steps'):
            return detector.prepare_geometry_steps(cur['points'], cur['img_metas'], ws_tag)
        return None

    def forward(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        losses = detector.forward_train(cur['points'], cur['img_metas'], cur['gt_bboxes_3d'],
                                        cur['gt_labels_3d'], cur.get('gt_bboxes_ignore', None))
        if self.weight != 1:
            losses = {k: v * self.weight for k, v in losses.items()}
        return _accumulate(ssl_obj, batch_dict, add_prefix(losses, self.name), prefer_sup=True)


# ------------------------------------------------------------------ box-list processors
class _MapBoxes(object):
    """in_bboxes_key -> out_bboxes_key, one sample at a time."""

    def _map(self, boxes, img_meta):
        raise NotImplementedError

    def forward(self, ssl_obj, batch_dict):
        metas = mlvl_get(batch_dict, self.img_metas)
        out = []
        for entry, meta in zip(mlvl_get(batch_dict, self.in_bboxes_key), metas):
            keep = None
            if isinstance(entry, Masked):            # a row map: filtered-out rows are mapped too and stay filtered out
                entry, keep = entry.full, entry.keep
            boxes, rest, tup = _split(entry)
            assert tup or isinstance(boxes, torch.Tensor)
            mapped = _join(self._map(boxes, meta), rest, tup)
            out.append(mapped if keep is None else Masked(mapped, keep))
        mlvl_set(batch_dict, self.out_bboxes_key, out)
        return batch_dict


@SSL_MODULES.register_module()
class BboxesTransform_3D(_MapBoxes):
    """processors_3d.py:12-56: apply (reverse=False) or undo (reverse=True) the 3D augmentations
    recorded in img_metas (flip / rot / scale / trans, in transformation_3d_flow order)."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, reverse, img_metas, in_bboxes_key, out_bboxes_key):
        self.reverse, self.img_metas = reverse, img_metas
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key

    def _map(self, boxes, img_meta):
        return apply_3d_transformation_bboxes(boxes, img_meta, reverse=self.reverse)


@SSL_MODULES.register_module()
class BboxesTransform_2D(_MapBoxes):
    """processors_2d.py:128-187: 'forward' (reverse=False) maps original-image boxes into the
    augmented image (ori2new), reverse maps back."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, reverse, img_metas, in_bboxes_key, out_bboxes_key):
        self.reverse, self.img_metas = reverse, img_metas
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key

    def _map(self, boxes, img_meta):
        assert boxes.shape[1] >= 4
        return bbox_2d_transform(img_meta, boxes, not self.reverse)


@SSL_MODULES.register_module()
class DetachBboxes(object):
    """processors_3d.py:59-78"""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, in_bboxes_key, out_bboxes_key):
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key

    def forward(self, ssl_obj, batch_dict):
        out = [Masked(tuple(t.detach() for t in entry.full), entry.keep) if isinstance(entry, Masked)
               else tuple(t.detach() for t in entry) for entry in mlvl_get(batch_dict, self.in_bboxes_key)]
        mlvl_set(batch_dict, self.out_bboxes_key, out)
        return batch_dict


@SSL_MODULES.register_module()
class Bboxes3DTo2D(object):
    """processors_3d.py:81-155: undo the 3D augs, project the 8 corners with lidar2img, take the
    clipped min/max box; optionally drop boxes that are invalid (behind the camera / empty)."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, img_metas='stu.img_metas', in_bboxes_key='stu.3d_bboxes_nms',
                 out_bboxes_key='stu.3d_bboxes_nms_2d_proj', filter_invalid=True):
        self.img_metas = img_metas
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key
        self.filter_invalid = filter_invalid

    def forward(self, ssl_obj, batch_dict):
        metas = mlvl_get(batch_dict, self.img_metas)
        out = []
        for entry, meta in zip(plain(mlvl_get(batch_dict, self.in_bboxes_key)), metas):
            boxes3d, rest, tup = _split(entry)
            if fused_on() and boxes3d.tensor.is_cuda and boxes3d.tensor.shape[1] == 7 and len(boxes3d.tensor):
                boxes2d, valid = unaug_project_boxes(boxes3d, meta)       # two launches, fwd + bwd
            else:
                boxes3d = apply_3d_transformation_bboxes(boxes3d, meta, reverse=True)
                boxes2d, valid = bbox_3d_to_bbox_2d(boxes3d, meta['lidar2img'], meta['ori_shape'])
            if self.filter_invalid and _lazy(boxes2d) and len(boxes2d):
                out.append(Masked(_join(boxes2d, rest, tup), valid))       # resolved by the consumer (NMS / matching)
                continue
            if self.filter_invalid:
                vi = valid.nonzero(as_tuple=False).squeeze(1)      # one mask -> index conversion
                boxes2d, rest = take(boxes2d, vi), [take(t, vi) for t in rest]
            out.append(_join(boxes2d, rest, tup))
        mlvl_set(batch_dict, self.out_bboxes_key, out)
        return batch_dict


@SSL_MODULES.register_module()
class MaxScoreFilter(object):
    """processors_fusion.py:9-47"""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, cls_includes_bg_pred, score_thr, in_bboxes_key, out_bboxes_key):
        self.cls_includes_bg_pred = cls_includes_bg_pred
        self.score_thr = score_thr
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key

    def forward(self, ssl_obj, batch_dict):
        out = []
        for entry in mlvl_get(batch_dict, self.in_bboxes_key):
            before = None
            if isinstance(entry, Masked):
                entry, before = entry.full, entry.keep
            scores = _fg_scores(entry[1], self.cls_includes_bg_pred)
            if len(scores) == 0:
                keep = torch.zeros((0,), dtype=torch.bool, device=scores.device)
            else:
                keep = scores.max(dim=1)[0] > self.score_thr
            if before is not None:
                keep = keep & before
            if _lazy(scores) and len(scores):
                out.append(Masked(tuple(entry), keep))
                continue
            ki = keep.nonzero(as_tuple=False).squeeze(1)
            out.append(tuple(take(t, ki) for t in entry))
        mlvl_set(batch_dict, self.out_bboxes_key, out)
        return batch_dict


@SSL_MODULES.register_module()
class FusionHungarianMatching(object):
    """processors_fusion.py:50-222: one-to-one match projected 3D boxes ('predictions') with 2D
    boxes ('GT') by cls + L1 + GIoU cost; matches dearer than cost_thr are dropped; outputs are
    index-aligned tuples.  Both score sets must be sigmoid probabilities."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, assigner_cfg, cost_thr, img_metas, cls_includes_bg_pred_3d,
                 cls_includes_bg_pred_2d, in_bboxes_3d_key, in_bboxes_2d_key, out_bboxes_3d_key,
                 out_bboxes_2d_key, match_cost_key=None, project_3d_to_2d=True):
        self.assigner = build_assigner(assigner_cfg)
        self.cost_thr = cost_thr
        self.img_metas = img_metas
        self.cls_includes_bg_pred_3d = cls_includes_bg_pred_3d
        self.cls_includes_bg_pred_2d = cls_includes_bg_pred_2d
        self.in_bboxes_3d_key, self.in_bboxes_2d_key = in_bboxes_3d_key, in_bboxes_2d_key
        self.out_bboxes_3d_key, self.out_bboxes_2d_key = out_bboxes_3d_key, out_bboxes_2d_key
        self.match_cost_key = match_cost_key
        self.project_3d_to_2d = project_3d_to_2d

    def _device_costs(self):
        """(w_cls, w_reg, w_iou, alpha, focal eps) when the assigner is the configured
        ModHungarianAssigner(DoubleSidedFocalLossCost, BBoxL1Cost xyxy, IoUCost giou), else None."""
        from .losses import BBoxL1Cost, DoubleSidedFocalLossCost, IoUCost, ModHungarianAssigner
        a = self.assigner
        if not (isinstance(a, ModHungarianAssigner) and isinstance(a.cls_cost, DoubleSidedFocalLossCost)
                and isinstance(a.reg_cost, BBoxL1Cost) and a.reg_cost.box_format == 'xyxy'
                and isinstance(a.iou_cost, IoUCost) and a.iou_cost.iou_mode == 'giou'
                and a.cls_cost.focal_loss_cost.gamma == 2):
            return None
        f = a.cls_cost.focal_loss_cost
        return float(f.weight), float(a.reg_cost.weight), float(a.iou_cost.weight), float(f.alpha), float(f.eps)

    def match_device(self, entry_3d, entry_2d, img_meta, costs_cfg):
        """match() with the whole cost matrix from ONE kernel (dm_fusion_match_cost), the LAP and the
        cost threshold on the host copy of it, and one upload of the matched index pairs."""
        from .. import _lib
        s3 = _fg_scores(entry_3d[1], self.cls_includes_bg_pred_3d).detach().float().contiguous()
        s2 = _fg_scores(entry_2d[1], self.cls_includes_bg_pred_2d).detach().float().contiguous()
        boxes2d = entry_2d[0].detach().float().contiguous()
        dev = boxes2d.device
        n3, n2, c = int(s3.shape[0]), int(s2.shape[0]), int(s3.shape[1])
        img_h, img_w, _ = img_meta['ori_shape']
        b3 = proj = m16 = None
        if self.project_3d_to_2d:
            b3 = entry_3d[0].tensor.detach().float().contiguous()
            m16 = _lib.floats(np.asarray(img_meta['lidar2img'].cpu() if torch.is_tensor(img_meta['lidar2img'])
                                         else img_meta['lidar2img'], np.float32).reshape(-1))
        else:
            proj = entry_3d[0].detach().float().contiguous()
        cost = torch.empty((n3, n2), dtype=torch.float32, device=dev)
        w_cls, w_reg, w_iou, alpha, feps = costs_cfg
        _lib.check(_lib.lib().dm_fusion_match_cost(
            _lib.ptr(b3), _lib.ptr(proj), _lib.ptr(s3), n3, _lib.ptr(boxes2d), _lib.ptr(s2), n2, c, m16,
            float(img_w), float(img_h), w_cls, w_reg, w_iou, alpha, feps, 1e-6, _lib.ptr(cost), None,
            _lib.stream()), 'dm_fusion_match_cost')
        host = cost.cpu().numpy()                                    # the one read-back of the module
        rows, cols = _lap_host(host)
        cm = host[rows, cols]
        keep = np.ones(len(rows), bool) if self.cost_thr is None else ~(cm > self.cost_thr)
        rows, cols, cm = rows[keep], cols[keep], cm[keep]
        packed = torch.from_numpy(np.stack([rows, cols]).astype(np.int64)).to(dev, non_blocking=True)
        return packed[0], packed[1], torch.from_numpy(cm.astype(np.float32)).to(dev, non_blocking=True)

    def match(self, entry_3d, entry_2d, img_meta, fused=None):
        """-> (index tensor into entry_3d, index tensor into entry_2d, matched costs)"""
        s3 = _fg_scores(entry_3d[1], self.cls_includes_bg_pred_3d)
        s2 = _fg_scores(entry_2d[1], self.cls_includes_bg_pred_2d)
        assert s3.shape[1] == s2.shape[1]
        boxes2d = entry_2d[0]
        cfg = self._device_costs() if fused_on(fused) else None
        if cfg is not None and s3.is_cuda and 0 < len(s3) <= 512 and 0 < len(s2) <= 512 and s3.shape[1] <= 8:
            return self.match_device(entry_3d, entry_2d, img_meta, cfg)
        if self.project_3d_to_2d:
            proj, _ = bbox_3d_to_bbox_2d(entry_3d[0], img_meta['lidar2img'], img_meta['ori_shape'])
        else:
            proj = entry_3d[0]
        img_h, img_w, _ = img_meta['ori_shape']
        factor = const([img_w, img_h, img_w, img_h], boxes2d.device, boxes2d.dtype).unsqueeze(0)
        proj_norm = bbox_xyxy_to_cxcywh(proj) / factor
        res = self.assigner.assign(proj_norm, torch.logit(s3, eps=1e-6), boxes2d,
                                   torch.logit(s2, eps=1e-6), dict(img_shape=img_meta['ori_shape']))
        gt_inds = res.gt_inds
        if res.max_overlaps is None:
            empty = gt_inds.new_zeros((0,))
            return empty, empty, proj.new_zeros((0,))
        matched = gt_inds > 0
        if self.cost_thr is not None:
            matched = matched & ~(res.max_overlaps > self.cost_thr)
        idx3 = matched.nonzero(as_tuple=True)[0]
        return idx3, gt_inds[idx3] - 1, res.max_overlaps[idx3]

    def _device_batch(self, e3s, e2s, metas, cfg):
        """All samples of the batch with ONE device->host copy: per sample one launch builds the whole cost matrix over
        the UNFILTERED rows (dm_fusion_match_cost), the matrices and the rows' filter masks come back together, the
        host solves each LAP on the rows / columns its masks keep (the cost of a pair does not depend on the other
        rows, so this is the matrix the compacted lists would give), applies the cost threshold, and one upload
        carries every matched index pair.  -> [(idx3, idx2, cost)] per sample, indices into the unfiltered rows."""
        from .. import _lib
        L = _lib.lib()
        w_cls, w_reg, w_iou, alpha, feps = cfg
        parts, shapes = [], []
        for e3, e2, meta in zip(e3s, e2s, metas):
            k3 = k2 = None
            if isinstance(e3, Masked):
                e3, k3 = e3.full, e3.keep
            if isinstance(e2, Masked):
                e2, k2 = e2.full, e2.keep
            s3 = _fg_scores(e3[1], self.cls_includes_bg_pred_3d).detach().float().contiguous()
            s2 = _fg_scores(e2[1], self.cls_includes_bg_pred_2d).detach().float().contiguous()
            boxes2d = e2[0].detach().float().contiguous()
            n3, n2, c = int(s3.shape[0]), int(s2.shape[0]), int(s3.shape[1])
            shapes.append((n3, n2, k3 is not None, k2 is not None))
            if n3 == 0 or n2 == 0:
                continue
            img_h, img_w, _ = meta['ori_shape']
            b3 = proj = m16 = None
            if self.project_3d_to_2d:
                b3 = e3[0].tensor.detach().float().contiguous()
                m16 = _lib.floats(np.asarray(meta['lidar2img'].cpu() if torch.is_tensor(meta['lidar2img'])
                                             else meta['lidar2img'], np.float32).reshape(-1))
            else:
                proj = e3[0].detach().float().contiguous()
            cost = torch.empty((n3, n2), dtype=torch.float32, device=boxes2d.device)
            _lib.check(L.dm_fusion_match_cost(
                _lib.ptr(b3), _lib.ptr(proj), _lib.ptr(s3), n3, _lib.ptr(boxes2d), _lib.ptr(s2), n2, c, m16,
                float(img_w), float(img_h), w_cls, w_reg, w_iou, alpha, feps, 1e-6, _lib.ptr(cost), None,
                _lib.stream()), 'dm_fusion_match_cost')
            parts.append(cost.view(-1))
            if k3 is not None:
                parts.append(k3.float())
            if k2 is not None:
                parts.append(k2.float())
        dev = None
        for e in list(e3s) + list(e2s):
            t = (e.full if isinstance(e, Masked) else e)[1]
            dev = t.device
            break
        host = torch.cat(parts).cpu().numpy() if parts else np.zeros((0,), np.float32)   # THE read-back of the module
        pos, found = 0, []
        for n3, n2, has3, has2 in shapes:
            if n3 == 0 or n2 == 0:
                found.append((np.zeros((0,), np.int64),) * 2 + (np.zeros((0,), np.float32),))
                continue
            cm = host[pos:pos + n3 * n2].reshape(n3, n2)
            pos += n3 * n2
            r = np.arange(n3)
            if has3:
                r = np.nonzero(host[pos:pos + n3] > 0.5)[0]
                pos += n3
            cc = np.arange(n2)
            if has2:
                cc = np.nonzero(host[pos:pos + n2] > 0.5)[0]
                pos += n2
            if len(r) == 0 or len(cc) == 0:
                found.append((np.zeros((0,), np.int64),) * 2 + (np.zeros((0,), np.float32),))
                continue
            sub = np.ascontiguousarray(cm[np.ix_(r, cc)])
            rows, cols = _lap_host(sub)
            val = sub[rows, cols]
            keep = np.ones(len(rows), bool) if self.cost_thr is None else ~(val > self.cost_thr)
            found.append((r[rows[keep]].astype(np.int64), cc[cols[keep]].astype(np.int64), val[keep].astype(np.float32)))
        # one upload for all samples: [idx3 | idx2] of every sample as int64, the matched costs as float32
        flat_i = np.concatenate([np.concatenate([a, b]) for a, b, _ in found]) if found else np.zeros((0,), np.int64)
        flat_c = np.concatenate([c for _, _, c in found]) if found else np.zeros((0,), np.float32)
        di = torch.from_numpy(flat_i).to(dev, non_blocking=True)
        dc = torch.from_numpy(flat_c).to(dev, non_blocking=True)
        out, pi, pc = [], 0, 0
        for a, b, c in found:
            k = len(a)
            out.append((di[pi:pi + k], di[pi + k:pi + 2 * k], dc[pc:pc + k]))
            pi += 2 * k
            pc += k
        return out

    def forward(self, ssl_obj, batch_dict):
        metas = mlvl_get(batch_dict, self.img_metas)
        e3s, e2s = mlvl_get(batch_dict, self.in_bboxes_3d_key), mlvl_get(batch_dict, self.in_bboxes_2d_key)
        out3, out2, costs = [], [], []
        cfg = self._device_costs() if fused_on() else None

        def ok(e3, e2):
            a = (e3.full if isinstance(e3, Masked) else e3)[1]
            b = (e2.full if isinstance(e2, Masked) else e2)[1]
            return a.is_cuda and len(a) <= 512 and len(b) <= 512 and a.dim() == 2 and \
                _fg_scores(a, self.cls_includes_bg_pred_3d).shape[1] <= 8
        if cfg is not None and len(e3s) and all(ok(a, b) for a, b in zip(e3s, e2s)):
            for (i3, i2, c), e3, e2 in zip(self._device_batch(e3s, e2s, metas, cfg), e3s, e2s):
                e3 = e3.full if isinstance(e3, Masked) else e3
                e2 = e2.full if isinstance(e2, Masked) else e2
                out3.append(tuple(take(t, i3) for t in e3))
                out2.append(tuple(take(t, i2) for t in e2))
                costs.append(c)
        else:
            for e3, e2, meta in zip(plain(e3s), plain(e2s), metas):
                i3, i2, c = self.match(e3, e2, meta)
                out3.append(tuple(take(t, i3) for t in e3))
                out2.append(tuple(take(t, i2) for t in e2))
                costs.append(c)
        mlvl_set(batch_dict, self.out_bboxes_3d_key, out3)
        mlvl_set(batch_dict, self.out_bboxes_2d_key, out2)
        if self.match_cost_key is not None:
            mlvl_set(batch_dict, self.match_cost_key, costs)
        return batch_dict


# ------------------------------------------------------------------ 2D modules
@SSL_MODULES.register_module()
class SimpleTest_2D(object):
    """processors_2d.py:11-86: Faster R-CNN test path up to (not including) NMS:
    (decoded boxes N x 4, softmax/sigmoid scores N x (C+1)), in the augmented image frame."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    hoistable = True          # reads only the raw batch (see Opd_SimpleTest_3D)
    early_before_geometry = True      # ... and no 3D geometry: SSL issues it while the geometry's read-backs are pending

    def __init__(self, ssl_obj_attr='teacher', batch_dict_key='tea', out_bboxes_key='2d_simple_test'):
        self.ssl_obj_attr = ssl_obj_attr
        self.batch_dict_key = batch_dict_key
        self.out_bboxes_key = out_bboxes_key

    def issue_early(self, ssl_obj, batch_dict):
        """Scheduling only (see Opd_SimpleTest_3D.issue_early): the pass has no read-back of its own — the NMS module
        that consumes it has —, so issuing it early is running it early."""
        self.forward(ssl_obj, batch_dict)

    def forward(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        if self.out_bboxes_key in cur:       # issued early
            return batch_dict
        cur[self.out_bboxes_key] = detector.simple_test_pre_nms(cur['img'], cur['img_metas'])
        return batch_dict


@SSL_MODULES.register_module()
class BboxesNMS_2D(object):
    """processors_2d.py:89-125: per-class NMS that carries the FULL score vector of each kept box."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, nms_cfg, cls_includes_bg_pred, batch_dict_key='stu',
                 in_bboxes_key='3d_bboxes_nms_2d_proj',
                 out_bboxes_key='3d_bboxes_nms_2d_proj_2d_nms'):
        self.cls_includes_bg_pred = cls_includes_bg_pred
        self.nms_cfg = nms_cfg
        self.batch_dict_key = batch_dict_key
        self.in_bboxes_key, self.out_bboxes_key = in_bboxes_key, out_bboxes_key

    def forward(self, ssl_obj, batch_dict):
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        entries = cur[self.in_bboxes_key]
        first = entries[0].full if entries and isinstance(entries[0], Masked) else (entries[0] if entries else None)
        if first is not None and _lazy(first[1]) and self.nms_cfg.get('nms_pre', -1) <= 0:
            # no compaction in front of the NMS (the score threshold and an incoming filter mask become "score below
            # every valid one"), none behind it: max_num rows + a mask for the consumer (bbox_utils.filter_by_nms_2d_masked)
            res = filter_by_nms_2d_masked([(e.full, e.keep) if isinstance(e, Masked) else (e, None) for e in entries],
                                          self.nms_cfg, not self.cls_includes_bg_pred)
            cur[self.out_bboxes_key] = [r if k is None else Masked(r, k) for r, k in res]
            return batch_dict
        cur[self.out_bboxes_key] = filter_by_nms_2d(plain(entries), self.nms_cfg, not self.cls_includes_bg_pred)
        return batch_dict


@SSL_MODULES.register_module()
class AverageBboxes_2D(object):
    """processors_2d.py:190-241: element-wise mean of two index-aligned box lists."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, cls_includes_bg_pred_1, cls_includes_bg_pred_2, in_bboxes_1_key,
                 in_bboxes_2_key, out_bboxes_key, out_score_type='averaged'):
        self.cls_includes_bg_pred_1 = cls_includes_bg_pred_1
        self.cls_includes_bg_pred_2 = cls_includes_bg_pred_2
        self.in_bboxes_1_key, self.in_bboxes_2_key = in_bboxes_1_key, in_bboxes_2_key
        self.out_bboxes_key = out_bboxes_key
        self.out_score_type = out_score_type

    def forward(self, ssl_obj, batch_dict):
        out = []
        for e1, e2 in zip(plain(mlvl_get(batch_dict, self.in_bboxes_1_key)),
                          plain(mlvl_get(batch_dict, self.in_bboxes_2_key))):
            boxes = (e1[0] + e2[0]) / 2
            if self.out_score_type == 'averaged':
                score = (_fg_scores(e1[1], self.cls_includes_bg_pred_1) +
                         _fg_scores(e2[1], self.cls_includes_bg_pred_2)) / 2
            elif self.out_score_type == 'pred_1':
                score = e1[1]
            elif self.out_score_type == 'pred_2':
                score = e2[1]
            else:
                raise ValueError(self.out_score_type)
            out.append((boxes, score))
        mlvl_set(batch_dict, self.out_bboxes_key, out)
        return batch_dict


@SSL_MODULES.register_module()
class TwoStageSupervised_2D(object):
    """consumers_2d.py:8-52"""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    # forward() adds losses and nothing else: no later module reads tensors of this pass's autograd
    # graph, so SSL.forward_train may back-propagate these losses as soon as they exist
    self_contained_losses = True

    def __init__(self, loss_detach_keys=[], ssl_obj_attr='student', batch_dict_key='stu'):
        self.loss_detach_keys = loss_detach_keys
        self.ssl_obj_attr = ssl_obj_attr
        self.batch_dict_key = batch_dict_key

    def forward(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        cur = mlvl_get(batch_dict, self.batch_dict_key)
        losses = detector.forward_train(cur['img'], cur['img_metas'], cur['gt_bboxes'],
                                        cur['gt_labels'], cur.get('gt_bboxes_ignore', None))
        for k in self.loss_detach_keys:
            losses.pop(k)
        return _accumulate(ssl_obj, batch_dict, add_prefix(losses, '%s' % self.batch_dict_key),
                           prefer_sup=True)


@SSL_MODULES.register_module()
class HardPseudoLabel_2D(object):
    """consumers_2d.py:55-121"""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    # forward() adds losses and nothing else: no later module reads tensors of this pass's autograd
    # graph, so SSL.forward_train may back-propagate these losses as soon as they exist
    self_contained_losses = True

    def __init__(self, score_thr, cls_includes_bg_pred, loss_detach_keys=[], ssl_obj_attr='student',
                 target_bboxes_key='tea.2d_bboxes_nms_stu_aug', target_img_key='stu.img',
                 target_img_metas_key='stu.img_metas', name='hard_pseudo_2d', weight=1):
        self.score_thr = score_thr
        self.cls_includes_bg_pred = cls_includes_bg_pred
        self.loss_detach_keys = loss_detach_keys
        self.ssl_obj_attr = ssl_obj_attr
        self.target_bboxes_key = target_bboxes_key
        self.target_img_key = target_img_key
        self.target_img_metas_key = target_img_metas_key
        self.weight = weight
        self.name = name

    def forward(self, ssl_obj, batch_dict):
        detector = mlvl_getattr(ssl_obj, self.ssl_obj_attr)
        boxes, labels = _threshold_pseudo(mlvl_get(batch_dict, self.target_bboxes_key),
                                          self.score_thr, self.cls_includes_bg_pred,
                                          lambda s: s.new_zeros((0, 4)))
        losses = detector.forward_train(mlvl_get(batch_dict, self.target_img_key),
                                        mlvl_get(batch_dict, self.target_img_metas_key),
                                        boxes, labels)
        losses = ssl_obj._collapse_losses(losses)
        for k in self.loss_detach_keys:
            losses.pop(k)
        for k in losses.keys():
            if 'acc' not in k:
                losses[k] = losses[k] * self.weight
        return _accumulate(ssl_obj, batch_dict, add_prefix(losses, self.name), prefer_sup=False)


# ------------------------------------------------------------------ consistency + metrics
class _FusedConsistencyLoss(torch.autograd.Function):
    """(cls, l1, iou) of one sample's matched pairs in one launch, analytic gradients in one more
    (csrc/consistency_loss.hip)."""

    @staticmethod
    def forward(ctx, in_boxes, in_scores, tgt_boxes, tgt_scores, img_w, img_h, alpha, gamma):
        from .. import _lib
        in_boxes, in_scores = in_boxes.detach().float().contiguous(), in_scores.detach().float().contiguous()
        tgt_boxes, tgt_scores = tgt_boxes.detach().float().contiguous(), tgt_scores.detach().float().contiguous()
        n, c = in_scores.shape
        dev = in_boxes.device
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        gs = torch.empty((n, c), dtype=torch.float32, device=dev)
        gl = torch.empty((n, 4), dtype=torch.float32, device=dev)
        gi = torch.empty((n, 4), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().dm_consistency_loss_forward(
            _lib.ptr(in_boxes), _lib.ptr(in_scores), _lib.ptr(tgt_boxes), _lib.ptr(tgt_scores), n, c,
            float(img_w), float(img_h), float(alpha), float(gamma), 1e-6, 1e-6, _lib.ptr(out), _lib.ptr(gs),
            _lib.ptr(gl), _lib.ptr(gi), _lib.stream()), 'dm_consistency_loss_forward')
        ctx.save_for_backward(gs, gl, gi)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _lib
        gs, gl, gi = ctx.saved_tensors
        n, c = gs.shape
        d_scores, d_boxes = torch.empty_like(gs), torch.empty_like(gl)
        _lib.check(_lib.lib().dm_consistency_loss_backward(
            _lib.ptr(g.contiguous().float()), _lib.ptr(gs), _lib.ptr(gl), _lib.ptr(gi), n, c,
            _lib.ptr(d_scores), _lib.ptr(d_boxes), _lib.stream()), 'dm_consistency_loss_backward')
        return d_boxes, d_scores, None, None, None, None, None, None


@SSL_MODULES.register_module()
class HungarianConsistency(object):
    """consumers_3d.py:11-117: box-level 2D<->3D consistency over index-aligned matched lists:
    class loss (MSE on probabilities or focal on logits), L1 on image-normalised boxes, GIoU."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, loss_cls_cfg=None, loss_iou_cfg=None, loss_l1_cfg=None,
                 loss_weights_cfg=dict(), in_bboxes_key='stu.3d_bboxes_nms_2d_proj',
                 target_bboxes_key='tea.2d_bboxes_nms_stu_aug_hung_dtch_bboxes',
                 cls_includes_bg_pred_in=True, cls_includes_bg_pred_target=True,
                 target_img_metas_key=None, name=None):
        self.loss_cls_cfg = loss_cls_cfg
        self.loss_cls = build_loss(loss_cls_cfg) if loss_cls_cfg is not None else None
        self.loss_iou = build_loss(loss_iou_cfg) if loss_iou_cfg is not None else None
        self.loss_l1 = build_loss(loss_l1_cfg) if loss_l1_cfg is not None else None
        self.loss_weights_cfg = loss_weights_cfg
        self.in_bboxes_key, self.target_bboxes_key = in_bboxes_key, target_bboxes_key
        self.cls_includes_bg_pred_in = cls_includes_bg_pred_in
        self.cls_includes_bg_pred_target = cls_includes_bg_pred_target
        self.target_img_metas_key = target_img_metas_key
        self.name = name

    def _fusable(self, boxes):
        """The DetMatch configuration (focal on logits + L1 + GIoU, all 'mean') on a CUDA device."""
        from .losses import GIoULoss, L1Loss
        return (fused_on() and boxes.is_cuda and isinstance(self.loss_cls, FocalLoss)
                and isinstance(self.loss_l1, L1Loss) and isinstance(self.loss_iou, GIoULoss)
                and all(l.reduction == 'mean' for l in (self.loss_cls, self.loss_l1, self.loss_iou))
                and self.loss_iou.eps == 1e-6)

    def forward(self, ssl_obj, batch_dict):
        in_list = plain(mlvl_get(batch_dict, self.in_bboxes_key))
        tgt_list = plain(mlvl_get(batch_dict, self.target_bboxes_key))
        metas = mlvl_get(batch_dict, self.target_img_metas_key)
        per_sample = dict()
        active = [(n, l) for n, l in (('cls_loss', self.loss_cls), ('l1_loss', self.loss_l1),
                                      ('iou_loss', self.loss_iou)) if l is not None]
        for idx, (cur_in, cur_tgt) in enumerate(zip(in_list, tgt_list)):
            in_boxes, tgt_boxes = cur_in[0], cur_tgt[0]
            if len(cur_in[1]) == 0 or len(cur_tgt[1]) == 0:
                continue
            in_scores = _fg_scores(cur_in[1], self.cls_includes_bg_pred_in)
            tgt_scores = _fg_scores(cur_tgt[1], self.cls_includes_bg_pred_target)
            assert in_scores.shape[1] == tgt_scores.shape[1] == 3
            # the kernel's backward feeds the INPUT boxes / scores only: a target that carries a graph
            # (e.g. student 2D against student 3D) takes the tensor losses
            if self._fusable(in_boxes) and not tgt_boxes.requires_grad and not tgt_scores.requires_grad:
                img_h, img_w, _ = metas[idx]['img_shape']
                vals = _FusedConsistencyLoss.apply(in_boxes, in_scores, tgt_boxes, tgt_scores, img_w, img_h,
                                                   self.loss_cls.alpha, self.loss_cls.gamma)
                for k, (lname, fn) in enumerate((('cls_loss', self.loss_cls), ('l1_loss', self.loss_l1),
                                                 ('iou_loss', self.loss_iou))):
                    per_sample.setdefault(lname, []).append(vals[k] if fn.loss_weight == 1.0
                                                            else vals[k] * fn.loss_weight)
                continue
            for lname, fn in active:
                if lname == 'cls_loss':
                    if isinstance(fn, MSELoss):
                        val = fn(in_scores, tgt_scores)
                    elif isinstance(fn, FocalLoss):
                        val = fn(torch.logit(in_scores, eps=1e-6), torch.argmax(tgt_scores, dim=1))
                    else:
                        raise Exception('Not Yet Implemented')
                elif lname == 'l1_loss':
                    img_h, img_w, _ = metas[idx]['img_shape']
                    factor = const([img_w, img_h, img_w, img_h], in_boxes.device, in_boxes.dtype).unsqueeze(0)
                    val = fn(in_boxes / factor, tgt_boxes / factor)
                else:
                    val = fn(in_boxes, tgt_boxes)
                per_sample.setdefault(lname, []).append(val)
        ref = in_list[0][0]
        losses = {k: sum(v) / len(v) for k, v in per_sample.items()}
        losses = ssl_obj._collapse_losses(losses)
        for k, w in self.loss_weights_cfg.items():
            if k in losses:
                losses[k] = losses[k] * w
            else:
                losses[k] = torch.zeros((), dtype=torch.float32, device=ref.device, requires_grad=True)
        return _accumulate(ssl_obj, batch_dict, add_prefix(losses, self.name), prefer_sup=False)


@SSL_MODULES.register_module()
class NumPreds(object):
    """consumers/metrics.py:9-24: mean number of boxes per sample, logged as a metric."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, bboxes_key, out_name):
        self.bboxes_key, self.out_name = bboxes_key, out_name

    def forward(self, ssl_obj, batch_dict):
        entries = plain(mlvl_get(batch_dict, self.bboxes_key))
        num = sum(e[0].shape[0] if isinstance(e, tuple) else e.shape[0] for e in entries) / len(entries)
        first = entries[0][0] if isinstance(entries[0], tuple) else entries[0]
        batch_dict['ssl_losses']['metrics.' + self.out_name] = torch.full(
            (), float(num), device=first.device, dtype=torch.float)
        return batch_dict


@SSL_MODULES.register_module()
class Vis3D(object):
    """consumers/visualize.py: debugging output only — outside the hot path, kept
    as a pass-through so configs that list it still build."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself

    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def forward(self, ssl_obj, batch_dict):
        return batch_dict


@SSL_MODULES.register_module()
class Vis2D_Kitti(Vis3D):
    """consumers/visualize.py (2D variant, confthr_frcnn recipe): pass-through, as Vis3D."""
    takes_masked = True       # reads box lists through plain() / handles Masked entries itself
