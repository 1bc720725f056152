"""3D part of the TS_SSL_Dataset pipelines on the device.

Host mirror of the reference transforms (same class names, constructor arguments and recorded
`img_metas` keys) — mmdet3d/datasets/pipelines/transforms_3d.py: RandomFlip3D :59-170,
GlobalRotScaleTrans :520-690, PointShuffle :695-720, ObjectRangeFilter :722-768, PointsRangeFilter
:770-810; mmdet3d/datasets/teacher_student_ssl_dataset.py:26-33 (shared pipeline once, then the
student and the teacher pipelines on copies).

The reference applies each transform to the whole point tensor on the host, sample by sample, in the
loader workers.  Here a transform only RECORDS what it would do into a `View3D` (and transforms the
handful of GT boxes directly); `augment_points` then executes all views of a batch — student and
teacher views of every sample — with one dm_points_augment call (csrc/augment.hip): every point is
read once per view, and the raw frame never leaves HBM.

Random draws follow the reference's order (flip, rotation angle, scale, translation) from a
`numpy.random.RandomState`-compatible generator; the point order of PointShuffle comes from a device
permutation.  No CPU fallback: `augment_points` raises for host tensors."""
import ctypes

import numpy as np
import torch

from . import _lib
from .devconst import upload

AUG_PARAMS = 24          # DM_AUG_PARAMS


class View3D(object):
    """What one pipeline does to one sample's points: flip -> rotate -> scale -> translate -> range
    filter -> shuffle (the order of the DetMatch pipelines, split_0.py:575-620), plus the img_metas
    record the SSL modules replay (`transformation_3d_flow`, `pcd_*`)."""

    def __init__(self, source=0, meta=None):
        self.source = source                     # index of the raw frame this view reads
        self.flip_h = self.flip_v = False
        self.rot = np.eye(3, dtype=np.float32)   # p' = p @ rot  (= the recorded pcd_rotation)
        self.scale = 1.0
        self.trans = np.zeros(3, dtype=np.float32)
        self.range = None
        self.shuffle = False
        self.meta = dict(meta or {})
        self.meta.setdefault('transformation_3d_flow', [])
        self._stage = 0                          # guards the fused kernel's fixed order

    def copy(self):
        """copy.deepcopy(data) of teacher_student_ssl_dataset.py:28."""
        v = View3D(self.source)
        v.flip_h, v.flip_v, v.rot, v.scale = self.flip_h, self.flip_v, self.rot.copy(), self.scale
        v.trans, v.range, v.shuffle, v._stage = self.trans.copy(), self.range, self.shuffle, self._stage
        v.meta = {k: (list(x) if isinstance(x, list) else x) for k, x in self.meta.items()}
        return v

    @classmethod
    def from_meta(cls, meta, source=0, point_cloud_range=None, shuffle=False):
        """Replay a recorded img_meta (the keys apply_3d_transformation consumes,
        mmdet3d/models/ssl_modules/bbox_utils.py:135-160) on raw points: flow entries 'HF' / 'VF' /
        'R' / 'S' / 'T' must appear in the fused order."""
        v = cls(source)
        v.meta = dict(meta)
        stage = {'HF': 1, 'VF': 1, 'R': 2, 'S': 2, 'T': 2}
        for op in meta.get('transformation_3d_flow', []):
            v._advance(stage[op], op)
            if op == 'HF':
                v.flip_h = bool(meta['pcd_horizontal_flip'])
            elif op == 'VF':
                v.flip_v = bool(meta['pcd_vertical_flip'])
            elif op == 'R':
                v.rot = np.asarray(meta['pcd_rotation'], dtype=np.float32)
            elif op == 'S':
                v.scale = float(meta['pcd_scale_factor'])
            else:
                v.trans = np.asarray(meta['pcd_trans'], dtype=np.float32)
        if point_cloud_range is not None:
            v.range = np.asarray(point_cloud_range, dtype=np.float32)
        v.shuffle = shuffle
        return v

    def _advance(self, stage, what):
        if stage < self._stage:
            raise ValueError('%s after a later transform: the fused kernel applies flip, rotation, scale, '
                             'translation, range filter, shuffle in this order' % what)
        self._stage = stage

    def params(self):
        p = np.zeros(AUG_PARAMS, dtype=np.float32)
        p[0], p[1] = float(self.flip_h), float(self.flip_v)
        p[2:11] = np.asarray(self.rot, dtype=np.float32).reshape(-1)
        p[11] = np.float32(self.scale)
        p[12:15] = self.trans
        r = self.range if self.range is not None else [-np.inf] * 3 + [np.inf] * 3
        p[15:21] = np.asarray(r, dtype=np.float32)
        return p


class RandomFlip3D(object):
    """transforms_3d.py:59-170 with sync_2d=True semantics for the 3D side: the horizontal BEV flip
    follows the sample's 2D `flip` flag (drawn here with probability flip_ratio_bev_horizontal when
    absent — mmdet's RandomFlip draw, un-vendored: parity unpinned); 'HF' / 'VF' are appended to the
    flow."""

    def __init__(self, sync_2d=True, flip_ratio_bev_horizontal=0.0, flip_ratio_bev_vertical=0.0, **kwargs):
        self.sync_2d = sync_2d
        self.flip_ratio = flip_ratio_bev_horizontal
        self.flip_ratio_bev_vertical = flip_ratio_bev_vertical

    def __call__(self, view, rng, boxes=None):
        m = view.meta
        view._advance(1, 'RandomFlip3D')
        if self.sync_2d:
            if 'flip' not in m:
                m['flip'] = bool(rng.rand() < self.flip_ratio)
            m['pcd_horizontal_flip'], m['pcd_vertical_flip'] = m['flip'], False
        else:
            if 'pcd_horizontal_flip' not in m:
                m['pcd_horizontal_flip'] = bool(rng.rand() < self.flip_ratio)
            if 'pcd_vertical_flip' not in m:
                m['pcd_vertical_flip'] = bool(rng.rand() < self.flip_ratio_bev_vertical)
        if m['pcd_horizontal_flip']:
            view.flip_h = not view.flip_h
            if boxes is not None:
                boxes.flip('horizontal')
            m['transformation_3d_flow'].append('HF')
        if m['pcd_vertical_flip']:
            view.flip_v = not view.flip_v
            if boxes is not None:
                boxes.flip('vertical')
            m['transformation_3d_flow'].append('VF')
        return view


class GlobalRotScaleTrans(object):
    """transforms_3d.py:520-690.  The recorded `pcd_rotation` is the matrix M with p' = p @ M: the
    box rotation matrix when the sample has GT boxes (:611-616), its transpose otherwise (:604-607)."""

    def __init__(self, rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05),
                 translation_std=(0, 0, 0), shift_height=False):
        if not isinstance(rot_range, (list, tuple, np.ndarray)):
            rot_range = [-rot_range, rot_range]
        if not isinstance(translation_std, (list, tuple, np.ndarray)):
            translation_std = [translation_std] * 3
        assert all(s >= 0 for s in translation_std), 'translation_std should be positive'
        assert not shift_height
        self.rot_range, self.scale_ratio_range = list(rot_range), list(scale_ratio_range)
        self.translation_std = list(translation_std)

    def __call__(self, view, rng, boxes=None):
        m = view.meta
        view._advance(2, 'GlobalRotScaleTrans')
        if not np.array_equal(view.rot, np.eye(3, dtype=np.float32)) or view.scale != 1.0:
            raise ValueError('one GlobalRotScaleTrans per pipeline')
        angle = rng.uniform(self.rot_range[0], self.rot_range[1])
        a32 = torch.tensor(angle, dtype=torch.float32)         # tensor.new_tensor(angle): fp32 sin / cos
        c, s = float(torch.cos(a32)), float(torch.sin(a32))
        if boxes is not None and len(boxes.tensor) != 0:
            boxes.rotate(angle)
            rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
        else:
            rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
            m['pcd_rotation_scalar'] = angle
        view.rot = rot
        m['pcd_rotation'] = rot
        if 'pcd_scale_factor' not in m:
            m['pcd_scale_factor'] = rng.uniform(self.scale_ratio_range[0], self.scale_ratio_range[1])
        view.scale = m['pcd_scale_factor']
        trans = rng.normal(scale=np.array(self.translation_std, dtype=np.float32), size=3).T
        view.trans = np.asarray(trans, dtype=np.float32)
        m['pcd_trans'] = trans
        if boxes is not None:
            boxes.scale(view.scale)
            boxes.translate(torch.as_tensor(view.trans, device=boxes.tensor.device))
        m['transformation_3d_flow'].extend(['R', 'S', 'T'])
        return view


class PointsRangeFilter(object):
    def __init__(self, point_cloud_range):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)

    def __call__(self, view, rng=None, boxes=None):
        view._advance(3, 'PointsRangeFilter')
        view.range = self.pcd_range
        return view


class PointShuffle(object):
    def __call__(self, view, rng=None, boxes=None):
        view._advance(4, 'PointShuffle')
        view.shuffle = True
        return view


class ObjectRangeFilter(object):
    """transforms_3d.py:722-768: keep the boxes whose BEV centre is inside the range, wrap yaw to
    [-pi, pi).  -> (boxes, labels)"""

    def __init__(self, point_cloud_range):
        self.pcd_range = np.array(point_cloud_range, dtype=np.float32)
        self.bev_range = self.pcd_range[[0, 1, 3, 4]]

    def filter(self, boxes, labels):
        bev = boxes.bev
        r = self.bev_range
        mask = (bev[:, 0] > r[0]) & (bev[:, 1] > r[1]) & (bev[:, 0] < r[2]) & (bev[:, 1] < r[3])
        boxes = boxes[mask]
        boxes.limit_yaw(offset=0.5, period=2 * np.pi)
        return boxes, labels[mask]


def augment_points(raw_points, views, generator=None):
    """Run every view's point pipeline in one fused pass.
    raw_points: list of (N_i, C) fp32 device tensors (the raw frames); views: list of View3D
    (view.source indexes raw_points).  -> list of (M_v, C) tensors, one per view, in view order.
    One host read-back (the per-view kept counts) at the end."""
    if not views:
        return []
    _lib.require_device(*raw_points)
    dev = raw_points[0].device
    c = raw_points[0].shape[1]
    assert all(p.dtype == torch.float32 and p.shape[1] == c for p in raw_points)
    lens = [int(p.shape[0]) for p in raw_points]
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = raw_points[0] if len(raw_points) == 1 else torch.cat(raw_points, dim=0)
    n_views = len(views)
    src_off = [int(starts[v.source]) for v in views]
    src_len = [lens[v.source] for v in views]
    dst_off = np.concatenate([[0], np.cumsum(src_len)]).astype(np.int64)
    total = int(dst_off[-1])
    params = upload(np.stack([v.params() for v in views]), dev)
    perm = None
    if any(v.shuffle for v in views):
        # one batched draw: random keys, argsort inside each view's slot segment
        keys = torch.rand(total, device=dev, generator=generator)
        seg = torch.repeat_interleave(torch.arange(n_views, device=dev),
                                      torch.as_tensor(src_len, device=dev), output_size=total)
        keys = torch.where(upload(np.array([v.shuffle for v in views]), dev, torch.bool)[seg], keys,
                           torch.arange(total, device=dev, dtype=torch.float32) / max(total, 1))
        order = torch.argsort(seg.double() * 2.0 + keys.double())
        perm = (order - upload(dst_off[:-1], dev, torch.int64)[seg]).int().contiguous()
    out = torch.empty((total, c), dtype=torch.float32, device=dev)
    counts = torch.empty(n_views, dtype=torch.int32, device=dev)
    L = _lib.lib()
    nbytes = L.dm_points_augment_workspace_bytes(n_views, _lib.ints(src_len))
    ws = _lib.workspace(nbytes, dev, 'augment')
    _lib.check(L.dm_points_augment(_lib.ptr(flat), c, n_views, _lib.ints(src_off), _lib.ints(src_len),
                                   _lib.ints(dst_off[:-1]), _lib.ptr(params), _lib.ptr(perm), _lib.ptr(out),
                                   _lib.ptr(counts), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                   _lib.stream()), 'dm_points_augment')
    kept = counts.tolist()
    return [out[int(dst_off[v]):int(dst_off[v]) + kept[v]] for v in range(n_views)]


class TSSSLPipeline3D(object):
    """TS_SSL_Dataset.__getitem__ (teacher_student_ssl_dataset.py:26-33) for the 3D modality of a
    whole batch: the shared transforms run once per sample, then the student / teacher transforms on
    copies of the recorded state; all point work happens in one augment_points call."""

    def __init__(self, shared, student, teacher, object_range=None):
        self.shared, self.student, self.teacher = list(shared), list(student), list(teacher)
        self.object_filter = ObjectRangeFilter(object_range) if object_range is not None else None

    def __call__(self, frames, rng, generator=None):
        """frames: list of dict(points (N,4) device tensor, [gt_bboxes_3d, gt_labels_3d], [meta]).
        -> (stu, tea): lists of dict(points, img_metas[, gt_bboxes_3d, gt_labels_3d])."""
        views, stu, tea = [], [], []
        for i, f in enumerate(frames):
            boxes = f.get('gt_bboxes_3d', None)
            boxes = boxes.clone() if boxes is not None else None
            base = View3D(i, f.get('meta', None))
            for t in self.shared:
                t(base, rng, boxes)
            sv, tv = base.copy(), base.copy()
            sboxes = boxes.clone() if boxes is not None else None
            for t in self.student:
                t(sv, rng, sboxes)
            for t in self.teacher:
                t(tv, rng, None)
            s = dict(img_metas=sv.meta)
            if sboxes is not None:
                labels = f['gt_labels_3d']
                if self.object_filter is not None:
                    sboxes, labels = self.object_filter.filter(sboxes, labels)
                s.update(gt_bboxes_3d=sboxes, gt_labels_3d=labels)
            stu.append(s)
            tea.append(dict(img_metas=tv.meta))
            views += [sv, tv]
        pts = augment_points([f['points'] for f in frames], views, generator)
        for i in range(len(frames)):
            stu[i]['points'], tea[i]['points'] = pts[2 * i], pts[2 * i + 1]
        return stu, tea
