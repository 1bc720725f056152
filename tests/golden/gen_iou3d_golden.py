"""Generates tests/golden/iou3d_ref.npz from the REFERENCE rotated-BEV-IoU CPU code compiled
here (oracle/build_ref.py:build_iou3d — thirdparty/Spconv-OpenPCDet/pcdet/ops/iou3d_nms/src/
iou3d_cpu.cpp:232 boxes_iou_bev_cpu, the host twin of the device `iou_bev` the NMS kernel uses,
iou3d_nms_kernel.cu:215-222).  Run in the build container only (needs /root/reference):

    python tests/golden/gen_iou3d_golden.py

Contents (inputs + reference outputs only):
  a, b, iou        160 x 120 pcdet boxes [x,y,z,dx,dy,dz,heading] drawn in clusters (so that many
                   pairs overlap, touch or nearly touch) and the reference IoU matrix;
  nms_boxes        400 boxes already sorted by descending score (clusters of near-duplicates, the
                   NMS workload), the reference's 400x400 IoU matrix `nms_iou`, and
  keep_<thr>       the keep list the reference's greedy pass (iou3d_nms.cpp:117-133: box i is kept
                   unless an earlier KEPT box j has iou_bev(j, i) > thr) gives on that matrix, for
                   thr 0.1 (post_processing), 0.7 / 0.8 (proposal layers), 0.01.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)


def clustered_boxes(rng, n, n_clusters, jitter):
    c = np.concatenate([rng.uniform(0, 60, (n_clusters, 1)), rng.uniform(-30, 30, (n_clusters, 1)),
                        rng.uniform(-1.5, -0.5, (n_clusters, 1))], 1)
    base_yaw = rng.uniform(-np.pi, np.pi, n_clusters)
    cls = rng.integers(0, 3, n_clusters)
    which = rng.integers(0, n_clusters, n)
    boxes = np.concatenate([c[which] + rng.normal(0, jitter, (n, 3)) * [1, 1, 0.1],
                            SIZES[cls[which]] * rng.uniform(0.85, 1.15, (n, 3)),
                            (base_yaw[which] + rng.normal(0, 0.15, n))[:, None]], 1)
    # exact duplicates, axis-aligned and touching pairs: the corner cases of the clipping code
    boxes[1] = boxes[0]
    boxes[2, 6] = 0.0
    boxes[3] = boxes[2]
    boxes[3, 0] += boxes[2, 3]            # shares an edge with box 2
    boxes[4] = boxes[2]
    boxes[4, 6] = np.pi / 2
    return boxes.astype(np.float32)


def ref_iou(ref, a, b):
    out = torch.zeros((len(a), len(b)), dtype=torch.float32)
    ref.boxes_iou_bev_cpu(torch.from_numpy(a).contiguous(), torch.from_numpy(b).contiguous(), out)
    return out.numpy()


def greedy_keep(iou, thr):
    keep = []
    for i in range(len(iou)):
        if not any(iou[j, i] > thr for j in keep):
            keep.append(i)
    return np.array(keep, np.int64)


def main():
    assert build_ref.build_iou3d() is not None, 'reference iou3d_cpu.cpp did not build'
    ref = build_ref.load_ref(build_ref.IOU3D_NAME)
    rng = np.random.default_rng(7)
    a = clustered_boxes(rng, 160, 25, 0.8)
    # b: perturbed copies of a's boxes (most rows of the matrix see several partial overlaps)
    b = (a[rng.integers(0, 160, 120)] + rng.normal(0, 1, (120, 7)) * [0.6, 0.6, 0.05, 0.1, 0.1, 0.05, 0.3]
         ).astype(np.float32)
    b[:5] = a[:5]
    out = dict(a=a, b=b, iou=ref_iou(ref, a, b))
    nb = clustered_boxes(rng, 400, 40, 0.35)
    m = ref_iou(ref, nb, nb)
    out['nms_boxes'] = nb
    out['nms_iou'] = m
    for thr in (0.01, 0.1, 0.7, 0.8):
        out['keep_%g' % thr] = greedy_keep(m, np.float32(thr))
        print('thr %g: keep %d of %d' % (thr, len(out['keep_%g' % thr]), len(nb)))
    print('pairs with IoU > 0:', int((out['iou'] > 0).sum()), 'of', out['iou'].size)
    np.savez_compressed(os.path.join(HERE, 'iou3d_ref.npz'), **out)
    print(os.path.getsize(os.path.join(HERE, 'iou3d_ref.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
