"""Generates tests/golden/kitti_eval.npz by running the REFERENCE's own KITTI evaluation
(mmdet3d/core/evaluation/kitti_utils/eval.py, loaded by file path; build container only) on a seeded
multi-image set, for the metrics that run without a GPU: 2D boxes ('bbox') and orientation ('aos').

numba is absent here: its `jit` decorators are replaced by the identity (the decorated functions are
plain Python/numpy loops), `prange` by `range`.  The BEV / 3D metrics call numba-CUDA kernels
(kitti_utils/rotate_iou.py) and cannot run; they are pinned by the reference's own known-answer tests
instead (tests/test_kitti_eval.py).  The fixture holds inputs + reference outputs only.

    python tests/golden/gen_kitti_eval_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/mmdet3d/core/evaluation/kitti_utils/eval.py'
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    nb = types.ModuleType('numba')

    def jit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f
    nb.jit, nb.prange = jit, range
    sys.modules['numba'] = nb
    spec = importlib.util.spec_from_file_location('ref_kitti_eval', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_annos(rng, n_images):
    names = np.array(['Car', 'Pedestrian', 'Cyclist', 'Van', 'Person_sitting', 'DontCare'])
    gts, dts = [], []
    for _ in range(n_images):
        g = int(rng.integers(0, 9))
        gn = names[rng.choice(6, g, p=[.4, .2, .15, .1, .05, .1])] if g else np.zeros(0, dtype='<U14')
        x1, y1 = rng.uniform(0, 1100, g), rng.uniform(100, 300, g)
        w, h = rng.uniform(15, 160, g), rng.uniform(12, 120, g)
        gbox = np.stack([x1, y1, x1 + w, y1 + h], 1).reshape(-1, 4)
        dc = gn == 'DontCare'
        gt = dict(name=gn, bbox=gbox, alpha=np.where(dc, -10.0, rng.uniform(-np.pi, np.pi, g)),
                  truncated=np.where(dc, -1.0, rng.choice([0.0, 0.1, 0.2, 0.4, 0.6], g)),
                  occluded=np.where(dc, -1, rng.integers(0, 4, g)))
        # detections: jittered copies of most non-DontCare GTs + false positives (some inside DontCare)
        keep = (~dc) & (rng.random(g) < 0.8)
        k = int(keep.sum())
        jitter = rng.normal(0, 1, (k, 4)) * (0.04 * np.stack([w, h, w, h], 1)[keep])
        dbox = gbox[keep] + jitter
        dname = gn[keep].copy()
        swap = rng.random(k) < 0.1
        dname[swap] = rng.choice(['Car', 'Pedestrian', 'Cyclist'], int(swap.sum()))
        dname = np.where(np.isin(dname, ['Van']), 'Car', np.where(dname == 'Person_sitting', 'Pedestrian', dname))
        dalpha = gt['alpha'][keep] + rng.normal(0, 0.4, k)
        f = int(rng.integers(0, 4))
        fx, fy = rng.uniform(0, 1100, f), rng.uniform(100, 300, f)
        fbox = np.stack([fx, fy, fx + rng.uniform(15, 120, f), fy + rng.uniform(12, 100, f)], 1).reshape(-1, 4)
        if dc.any() and f:
            fbox[0] = gbox[dc][0] + [2, 2, -2, -2]
        dt = dict(name=np.concatenate([dname, rng.choice(['Car', 'Pedestrian', 'Cyclist'], f)]).astype('<U14'),
                  bbox=np.concatenate([dbox, fbox], 0).reshape(-1, 4),
                  alpha=np.concatenate([dalpha, rng.uniform(-np.pi, np.pi, f)]),
                  score=np.round(rng.uniform(0.05, 1.0, k + f), 3),
                  truncated=np.zeros(k + f), occluded=np.zeros(k + f, dtype=np.int64))
        gts.append(gt)
        dts.append(dt)
    return gts, dts


def main():
    ref = load_reference()
    rng = np.random.default_rng(2024)
    gts, dts = make_annos(rng, 60)
    min_overlaps = np.array([[[0.7, 0.5, 0.5], [0.7, 0.5, 0.5], [0.7, 0.5, 0.5]],
                             [[0.5, 0.25, 0.25], [0.5, 0.25, 0.25], [0.5, 0.25, 0.25]]])
    classes = [0, 1, 2]
    ret = ref.eval_class(gts, dts, classes, [0, 1, 2], 0, min_overlaps, True, 7)
    out = dict(n_images=np.array(len(gts)), min_overlaps=min_overlaps,
               recall=ret['recall'], precision=ret['precision'], orientation=ret['orientation'],
               mAP_bbox=ref.get_mAP(ret['precision']), mAP_aos=ref.get_mAP(ret['orientation']))
    for i, (g, d) in enumerate(zip(gts, dts)):
        for k, v in g.items():
            out['gt%d_%s' % (i, k)] = v
        for k, v in d.items():
            out['dt%d_%s' % (i, k)] = v
    # clean_data and thresholds on their own
    cd = [ref.clean_data(gts[i], dts[i], c, dif) for i in range(8) for c in classes for dif in range(3)]
    out['clean_num_valid'] = np.array([c[0] for c in cd])
    out['clean_ignored_gt'] = np.concatenate([np.array(c[1], dtype=np.int64) for c in cd])
    out['clean_ignored_dt'] = np.concatenate([np.array(c[2], dtype=np.int64) for c in cd])
    out['clean_num_dc'] = np.array([len(c[3]) for c in cd])
    sc = np.round(rng.uniform(0, 1, 57), 3)
    out['thr_scores'] = sc
    out['thr_out'] = np.array(ref.get_thresholds(sc.copy(), 80))
    np.savez_compressed(os.path.join(HERE, 'kitti_eval.npz'), **out)
    print('wrote kitti_eval.npz: %d images, mAP_bbox[class, difficulty, overlap] =\n%s' % (len(gts), out['mAP_bbox']))


if __name__ == '__main__':
    main()
