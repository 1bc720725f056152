"""KITTI detection AP — host mirror of mmdet3d/core/evaluation/kitti_utils/eval.py
(`kitti_eval`, `do_eval`, `eval_class`: same arguments, result string and `KITTI/...` keys).

What runs where:
  * rotated BEV / 3D overlaps: the device kernel of iou3d_nms.hip (`boxes_overlap_bev`) instead of the
    reference's numba-CUDA `rotate_iou_gpu_eval` (kitti_utils/rotate_iou.py); camera-frame boxes
    (x, z, l, w, ry).  rotate_iou.py turns its corners CLOCKWISE by the angle (:205-227), the device kernel
    counter-clockwise by the heading: the boxes are handed over with heading = -ry.
    No CPU path: BEV / 3D metrics raise without the GPU;
  * the greedy per-image matching (numba `compute_statistics_jit` / `fused_compute_statistics`,
    eval.py:161-338): dm_kitti_tp_scores_host / dm_kitti_pr_host, one call per (class, difficulty,
    min_overlap) cell instead of one Python->JIT call per image and score threshold;
  * 2D overlaps, `clean_data`, thresholds, precision/recall envelopes, mAP over 40 recall points:
    numpy, vectorised.
Pinned by the reference's own known-answer tests (tests/test_metrics/test_kitti_eval.py,
tests/test_data/test_datasets/test_kitti_dataset.py::test_evaluate) and by goldens generated from the
reference's eval.py for the 2D / AOS metrics (tests/golden/gen_kitti_eval_golden.py)."""
import ctypes

import numpy as np
import torch

from . import _lib

CLASS_NAMES = ['car', 'pedestrian', 'cyclist']
MIN_HEIGHT = [40, 25, 25]
MAX_OCCLUSION = [0, 1, 2]
MAX_TRUNCATION = [0.15, 0.3, 0.5]
N_SAMPLE_PTS = 41


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """eval.py:7-25: the score at which each of the 41 recall sample points is first reached."""
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    current_recall = 0
    thresholds = []
    n = len(scores)
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < n - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < n - 1:
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """eval.py:28-80, vectorised -> (num_valid_gt, ignored_gt, ignored_dt, dc_bboxes)."""
    cls = CLASS_NAMES[current_class]
    gt_names = np.char.lower(np.asarray(gt_anno['name'], dtype=str)) if len(gt_anno['name']) else np.zeros(0, str)
    gt_bbox = np.asarray(gt_anno['bbox'], dtype=np.float64).reshape(-1, 4)
    height = gt_bbox[:, 3] - gt_bbox[:, 1]
    same = gt_names == cls
    neighbour = ((cls == 'pedestrian') & (gt_names == 'person_sitting')) | ((cls == 'car') & (gt_names == 'van'))
    hard = (np.asarray(gt_anno['occluded'])[:len(gt_names)] > MAX_OCCLUSION[difficulty]) | \
        (np.asarray(gt_anno['truncated'])[:len(gt_names)] > MAX_TRUNCATION[difficulty]) | \
        (height <= MIN_HEIGHT[difficulty])
    ignored_gt = np.full(len(gt_names), -1, dtype=np.int64)
    ignored_gt[neighbour | (same & hard)] = 1
    ignored_gt[same & ~hard] = 0
    dc = gt_bbox[np.asarray(gt_anno['name'], dtype=str) == 'DontCare'] if len(gt_names) else np.zeros((0, 4))
    dt_names = np.char.lower(np.asarray(dt_anno['name'], dtype=str)) if len(dt_anno['name']) else np.zeros(0, str)
    dt_bbox = np.asarray(dt_anno['bbox'], dtype=np.float64).reshape(-1, 4)
    dt_h = np.abs(dt_bbox[:, 3] - dt_bbox[:, 1])
    ignored_dt = np.where(dt_h < MIN_HEIGHT[difficulty], 1, np.where(dt_names == cls, 0, -1)).astype(np.int64)
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc.astype(np.float64).reshape(-1, 4)


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """eval.py:83-112 (N,4) x (K,4) -> (N,K); criterion -1 IoU, 0 / boxes area, 1 / query area."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    q = np.asarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    iw = np.minimum(boxes[:, None, 2], q[None, :, 2]) - np.maximum(boxes[:, None, 0], q[None, :, 0])
    ih = np.minimum(boxes[:, None, 3], q[None, :, 3]) - np.maximum(boxes[:, None, 1], q[None, :, 1])
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.0)
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))[:, None]
    qarea = ((q[:, 2] - q[:, 0]) * (q[:, 3] - q[:, 1]))[None, :]
    ua = {-1: area + qarea - inter, 0: area + 0 * qarea, 1: qarea + 0 * area}.get(criterion, np.ones_like(inter))
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inter > 0, inter / ua, 0.0)


def _bev_intersection(boxes, qboxes):
    """Rotated overlap AREA of camera-frame BEV boxes (N,5) x (K,5) [x, z, l, w, ry] on the device."""
    if not torch.cuda.is_available():
        raise _lib.DetMatchHipError('the BEV / 3D KITTI metrics need the MI355X (rotated overlap kernel); '
                                    'there is no CPU path')
    from . import iou3d_nms
    dev = torch.device('cuda', torch.cuda.current_device())

    def as7(b):
        b = torch.as_tensor(np.asarray(b, dtype=np.float32), device=dev)
        z = torch.zeros_like(b[:, :1])
        return torch.cat([b[:, 0:2], z, b[:, 2:4], z + 1, -b[:, 4:5]], dim=1).contiguous()
    if len(boxes) == 0 or len(qboxes) == 0:
        return np.zeros((len(boxes), len(qboxes)), dtype=np.float64)
    # exact convex clipping (no 1 cm corner margin of the NMS kernel): matches near the 0.7 / 0.5 / 0.25
    # thresholds must not flip between TP and FP
    return iou3d_nms.boxes_overlap_bev_exact(as7(boxes), as7(qboxes)).double().cpu().numpy()


def bev_box_overlap(boxes, qboxes, criterion=-1):
    """eval.py:115-118 / rotate_iou.py: rotated IoU of (N,5) x (K,5) [x, z, l, w, ry]."""
    boxes, qboxes = np.asarray(boxes, dtype=np.float64), np.asarray(qboxes, dtype=np.float64)
    inter = _bev_intersection(boxes, qboxes)
    a1 = (boxes[:, 2] * boxes[:, 3])[:, None]
    a2 = (qboxes[:, 2] * qboxes[:, 3])[None, :]
    ua = {-1: a1 + a2 - inter, 0: a1 + 0 * a2, 1: a2 + 0 * a1}.get(criterion, np.ones_like(inter))
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inter > 0, inter / ua, 0.0)


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """eval.py:121-158: (N,7) x (K,7) camera boxes [x, y(bottom), z, l, h, w, ry] -> 3D IoU."""
    boxes, qboxes = np.asarray(boxes, dtype=np.float64), np.asarray(qboxes, dtype=np.float64)
    rinc = _bev_intersection(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]])
    ih = np.minimum(boxes[:, None, 1], qboxes[None, :, 1]) - \
        np.maximum((boxes[:, 1] - boxes[:, 4])[:, None], (qboxes[:, 1] - qboxes[:, 4])[None, :])
    inc = np.where((rinc > 0) & (ih > 0), ih * rinc, 0.0)
    v1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    v2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    ua = {-1: v1 + v2 - inc, 0: v1 + 0 * v2, 1: v2 + 0 * v1}.get(criterion, None)
    if ua is None:
        return np.where(inc > 0, 1.0, 0.0)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(inc > 0, inc / ua, 0.0)


def _cat(annos, key, width):
    """Concatenate a per-image field as (sum n, width) float64 (empty images contribute no rows)."""
    parts = [np.asarray(a[key], dtype=np.float64).reshape(-1, width) for a in annos]
    return np.concatenate(parts, 0) if parts else np.zeros((0, width))


def calculate_overlaps(dt_annos, gt_annos, metric):
    """eval.py:341-416 with the roles eval_class uses: per image a (num_dt, num_gt) matrix.
    All images go through ONE batched overlap call; the per-image blocks are cut from it."""
    n_dt = np.array([len(a['name']) for a in dt_annos], dtype=np.int64)
    n_gt = np.array([len(a['name']) for a in gt_annos], dtype=np.int64)

    def boxes(annos):
        if metric == 0:
            return _cat(annos, 'bbox', 4)
        loc, dims, rot = _cat(annos, 'location', 3), _cat(annos, 'dimensions', 3), _cat(annos, 'rotation_y', 1)
        if metric == 1:
            return np.concatenate([loc[:, [0, 2]], dims[:, [0, 2]], rot], 1)
        return np.concatenate([loc, dims, rot], 1)
    fn = {0: image_box_overlap, 1: bev_box_overlap, 2: d3_box_overlap}[metric]
    out, step = [], 256                       # images per batched call: bounds the (sum dt) x (sum gt) matrix
    for s in range(0, len(dt_annos), step):
        d, g = boxes(dt_annos[s:s + step]), boxes(gt_annos[s:s + step])
        full = fn(d, g).astype(np.float64)
        do = np.concatenate([[0], np.cumsum(n_dt[s:s + step])])
        go = np.concatenate([[0], np.cumsum(n_gt[s:s + step])])
        for i in range(len(do) - 1):
            out.append(np.ascontiguousarray(full[do[i]:do[i + 1], go[i]:go[i + 1]]))
    return out, n_dt, n_gt


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False,
               num_parts=200):
    """eval.py:450-569 -> dict(recall, precision, orientation), each
    [num_class, num_difficulty, num_minoverlap, 41]."""
    assert len(gt_annos) == len(dt_annos)
    L = _lib.lib()
    overlaps, n_dt, n_gt = calculate_overlaps(dt_annos, gt_annos, metric)
    ov_flat = np.concatenate([o.reshape(-1) for o in overlaps]) if overlaps else np.zeros(0)
    ov_flat = np.ascontiguousarray(ov_flat, dtype=np.float64)
    min_overlaps = np.asarray(min_overlaps, dtype=np.float64)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    gt_datas = np.ascontiguousarray(np.concatenate(
        [np.concatenate([np.asarray(a['bbox'], np.float64).reshape(-1, 4),
                         np.asarray(a['alpha'], np.float64).reshape(-1, 1)], 1) for a in gt_annos], 0))
    dt_datas = np.ascontiguousarray(np.concatenate(
        [np.concatenate([np.asarray(a['bbox'], np.float64).reshape(-1, 4),
                         np.asarray(a['alpha'], np.float64).reshape(-1, 1),
                         np.asarray(a['score'], np.float64).reshape(-1, 1)], 1) for a in dt_annos], 0))
    for m, current_class in enumerate(current_classes):
        for idx_l, difficulty in enumerate(difficultys):
            cleaned = [clean_data(g, d, current_class, difficulty) for g, d in zip(gt_annos, dt_annos)]
            total_valid = sum(c[0] for c in cleaned)
            ig = np.ascontiguousarray(np.concatenate([c[1] for c in cleaned]), dtype=np.int64)
            idt = np.ascontiguousarray(np.concatenate([c[2] for c in cleaned]), dtype=np.int64)
            dcs = np.ascontiguousarray(np.concatenate([c[3] for c in cleaned], 0), dtype=np.float64)
            n_dc = np.array([len(c[3]) for c in cleaned], dtype=np.int64)
            common = (_ptr(ov_flat), _ptr(n_gt), _ptr(n_dt), _ptr(n_dc), len(gt_annos), _ptr(gt_datas),
                      _ptr(dt_datas), _ptr(dcs), _ptr(ig), _ptr(idt), int(metric))
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = np.zeros(max(int(n_gt.sum()), 1), dtype=np.float64)
                cnt = L.dm_kitti_tp_scores_host(*common, float(min_overlap), _ptr(scores))
                if cnt < 0:
                    raise _lib.DetMatchHipError('dm_kitti_tp_scores_host: invalid input')
                thresholds = np.ascontiguousarray(get_thresholds(scores[:cnt], total_valid), dtype=np.float64)
                pr = np.zeros((len(thresholds), 4), dtype=np.float64)
                _lib.check(L.dm_kitti_pr_host(*common, float(min_overlap), _ptr(thresholds), len(thresholds),
                                              int(compute_aos), _ptr(pr)), 'dm_kitti_pr_host')
                n = len(thresholds)
                with np.errstate(divide='ignore', invalid='ignore'):
                    recall[m, idx_l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, idx_l, k, :n] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, idx_l, k, :n] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                # monotone envelopes over the WHOLE 41-slot row (eval.py:558-565: max over [i:])
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    row = arr[m, idx_l, k]
                    env = np.maximum.accumulate(row[::-1])[::-1]
                    row[:n] = env[:n]
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def get_mAP(prec):
    """eval.py:578-582: mean over recall points 1..40, in percent."""
    return prec[..., 1:].sum(-1) / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types=('bbox', 'bev', '3d')):
    """eval.py:594-628 -> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos), each [num_class, 3, num_minoverlap]."""
    difficultys = [0, 1, 2]
    mAP_bbox = mAP_aos = mAP_bev = mAP_3d = None
    if 'bbox' in eval_types:
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps,
                         compute_aos=('aos' in eval_types))
        mAP_bbox = get_mAP(ret['precision'])
        if 'aos' in eval_types:
            mAP_aos = get_mAP(ret['orientation'])
    if 'bev' in eval_types:
        mAP_bev = get_mAP(eval_class(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps)['precision'])
    if '3d' in eval_types:
        mAP_3d = get_mAP(eval_class(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps)['precision'])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


_CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=('bbox', 'bev', '3d')):
    """eval.py:650-781 -> (result string, dict of `KITTI/<Class>_<3D|BEV|2D>_<difficulty>_<strict|loose>`
    and `KITTI/Overall_*` values)."""
    eval_types = list(eval_types)
    assert len(eval_types) > 0, 'must contain at least one evaluation type'
    if 'aos' in eval_types:
        assert 'bbox' in eval_types, 'must evaluate bbox when evaluating aos'
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3)
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])
    min_overlaps = np.stack([overlap_0_7, overlap_0_5], axis=0)
    name_to_class = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    current_classes = [name_to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = min_overlaps[:, :, current_classes]
    pred_alpha = any((np.asarray(a['alpha']) != -10).any() for a in dt_annos if len(a['alpha']))
    valid_alpha_gt = any(len(a['alpha']) != 0 and a['alpha'][0] != -10 for a in gt_annos)
    compute_aos = bool(pred_alpha and valid_alpha_gt)
    if compute_aos and 'aos' not in eval_types:
        eval_types.append('aos')
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types)
    result, ret_dict = '', {}
    difficulty = ['easy', 'moderate', 'hard']
    for j, curcls in enumerate(current_classes):
        name = _CLASS_TO_NAME[curcls]
        for i in range(min_overlaps.shape[0]):
            result += '{} AP@{:.2f}, {:.2f}, {:.2f}:\n'.format(name, *min_overlaps[i, :, j])
            if mAPbbox is not None:
                result += 'bbox AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAPbbox[j, :, i])
            if mAPbev is not None:
                result += 'bev  AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAPbev[j, :, i])
            if mAP3d is not None:
                result += '3d   AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAP3d[j, :, i])
            if compute_aos:
                result += 'aos  AP:{:.2f}, {:.2f}, {:.2f}\n'.format(*mAPaos[j, :, i])
            for idx in range(3):
                postfix = '%s_%s' % (difficulty[idx], 'strict' if i == 0 else 'loose')
                prefix = 'KITTI/%s' % name
                if mAP3d is not None:
                    ret_dict['%s_3D_%s' % (prefix, postfix)] = mAP3d[j, idx, i]
                if mAPbev is not None:
                    ret_dict['%s_BEV_%s' % (prefix, postfix)] = mAPbev[j, idx, i]
                if mAPbbox is not None:
                    ret_dict['%s_2D_%s' % (prefix, postfix)] = mAPbbox[j, idx, i]
    if len(current_classes) > 1:
        result += '\nOverall AP@{}, {}, {}:\n'.format(*difficulty)
        if mAPbbox is not None:
            mAPbbox = mAPbbox.mean(axis=0)
            result += 'bbox AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAPbbox[:, 0])
        if mAPbev is not None:
            mAPbev = mAPbev.mean(axis=0)
            result += 'bev  AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAPbev[:, 0])
        if mAP3d is not None:
            mAP3d = mAP3d.mean(axis=0)
            result += '3d   AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*mAP3d[:, 0])
        if compute_aos:
            mAPaos = mAPaos.mean(axis=0)
            result += 'aos  AP:{:.2f}, {:.2f}, {:.2f}\n'.format(*mAPaos[:, 0])
        for idx in range(3):
            if mAP3d is not None:
                ret_dict['KITTI/Overall_3D_%s' % difficulty[idx]] = mAP3d[idx, 0]
            if mAPbev is not None:
                ret_dict['KITTI/Overall_BEV_%s' % difficulty[idx]] = mAPbev[idx, 0]
            if mAPbbox is not None:
                ret_dict['KITTI/Overall_2D_%s' % difficulty[idx]] = mAPbbox[idx, 0]
    return result, ret_dict
