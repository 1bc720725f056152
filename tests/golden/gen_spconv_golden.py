"""Generates tests/golden/spconv_ref_*.npz from the REFERENCE sparse-conv CPU code compiled
here (oracle/build_ref.py:build_spconv — the reference's unpatched `spconv/geometry.h`
getIndicePairsConv :145 / getIndicePairsSubM :248 and `src/reordering.cc` gather/scatter-add
functors :21-50, driven through the call sequence of `spconv_ops.h:28-141,260-456`).
Run in the build container only (needs /root/reference):

    python tests/golden/gen_spconv_golden.py

Fixtures hold inputs + the reference's outputs, nothing else:

  spconv_ref_small.npz   two spatial crops (≈1.2 m windows around a cuboid) of the synthetic
        KITTI-shaped frames 0 and 1, voxelised by the reference voxeliser, B = 2, chained
        through the 12 layers / 8 rulebooks of VoxelBackBone8x (spconv_backbone.py:80-120) at
        the real channel widths with seeded N(0, 0.05^2) weights (`layer_weight(li, ...)`, not
        stored: regenerated from the seed) and ReLU between layers:
        every rulebook in the reference's FIRST-TOUCH order (outids, the valid part of
        indicePairs, indiceNum), every layer's forward output, input gradient and weight
        gradient (dy = seeded N(0,1), `layer_dy`).  Weight gradients of the layers with >= 32x32
        weights keep the taps 0 / centre / last only (size).
  spconv_ref_full.npz    the full frames 0 and 1 (B = 2, ≈29 k voxels): per rulebook
        indiceNum, N_out and SHA-1 digests of (a) the reference's raw outids / indicePairs
        bytes (first-touch order: pins the oracle bit for bit) and (b) an order-free canonical
        form (pins implementations with another, documented, output order — the HIP rulebook
        sorts strided outputs by cell id): per offset the pairs as (input CELL id, output CELL id)
        sorted by input cell, and the sorted output cell ids.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from detmatch_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPE = [41, 1600, 1408]
LAYERS = [   # (rulebook key, subm, cin, cout, ksize, stride, padding)  spconv_backbone.py:80-120
    ('subm1', True, 4, 16, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('subm1', True, 16, 16, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('spconv2', False, 16, 32, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
    ('subm2', True, 32, 32, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('subm2', True, 32, 32, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('spconv3', False, 32, 64, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
    ('subm3', True, 64, 64, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('subm3', True, 64, 64, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('spconv4', False, 64, 64, [3, 3, 3], [2, 2, 2], [0, 1, 1]),
    ('subm4', True, 64, 64, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('subm4', True, 64, 64, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ('spconv_down2', False, 64, 128, [3, 1, 1], [2, 1, 1], [0, 0, 0]),
]


def layer_weight(li, ks, cin, cout):
    return (np.random.default_rng(1000 + li).standard_normal(list(ks) + [cin, cout]) * 0.05
            ).astype(np.float32)


def layer_dy(li, shape):
    return np.random.default_rng(2000 + li).standard_normal(shape).astype(np.float32)


def out_size(shape, ks, st, pd):
    """mmdet3d/ops/spconv/ops.py:20-31 (dilation 1)"""
    return [(shape[i] + 2 * pd[i] - (ks[i] - 1) - 1) // st[i] + 1 for i in range(3)]


def voxelize(frames):
    feats, coors = [], []
    for b, pts in enumerate(frames):
        v, c, n = build_ref.ref_hard_voxelize(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        feats.append((v.sum(1) / np.maximum(n, 1)[:, None]).astype(np.float32))
        coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
    return np.concatenate(feats), np.concatenate(coors).astype(np.int32)


def crop(frame, half=0.8):
    """Points within a square window around the cuboid nearest to the sensor."""
    pts, gt = frame['points'], frame['gt_boxes']
    c = gt[np.argmin(np.hypot(gt[:, 0], gt[:, 1]))]
    m = (np.abs(pts[:, 0] - c[0]) < half) & (np.abs(pts[:, 1] - c[1]) < half)
    return pts[m]


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def cell_id(ids, shape):
    ids = ids.astype(np.int64)
    return ((ids[:, 0] * shape[0] + ids[:, 1]) * shape[1] + ids[:, 2]) * shape[2] + ids[:, 3]


def canonical(inids, in_shape, outids, pairs, num, out_shape):
    """Order-free digests: per offset (input CELL id, output CELL id) sorted by input cell."""
    cells = cell_id(outids, out_shape)
    in_cells = cell_id(inids, in_shape)
    h = hashlib.sha1()
    for k in range(pairs.shape[0]):
        i = in_cells[pairs[k, 0, :num[k]]]
        o = cells[pairs[k, 1, :num[k]]]
        order = np.lexsort((o, i))
        h.update(np.ascontiguousarray(np.stack([i[order], o[order]], 1)).tobytes())
    return h.hexdigest(), sha(np.sort(cells))


def rulebooks(ref, idx, batch):
    books, shape, cur = {}, SHAPE, torch.from_numpy(idx)
    for key, subm, cin, cout, ks, st, pd in LAYERS:
        if key in books:
            continue
        osh = shape if subm else out_size(shape, ks, st, pd)
        o, p, n = ref.get_indice_pairs(cur, batch, osh, ks, st, pd, [1, 1, 1], subm)
        books[key] = (cur, o, p, n, shape, osh)
        cur, shape = o, osh
    return books


def main():
    build_ref.build()
    assert build_ref.build_spconv() is not None, 'reference spconv sources did not build'
    ref = build_ref.load_ref(build_ref.SPCONV_NAME)
    frames = [synth.lidar_frame(s) for s in (0, 1)]

    # ---------------- small: full data ----------------
    feats, idx = voxelize([crop(f) for f in frames])
    books = rulebooks(ref, idx, 2)
    out = dict(features=feats, indices=idx, spatial_shape=np.array(SHAPE))
    for key, (inp, o, p, n, ish, osh) in books.items():
        nn = n.numpy()
        out['rb_%s_outids' % key] = o.numpy()
        out['rb_%s_num' % key] = nn
        out['rb_%s_out_shape' % key] = np.array(osh)
        # valid part of indicePairs[k, :, :num[k]], offsets concatenated
        out['rb_%s_pairs_in' % key] = np.concatenate([p[k, 0, :nn[k]].numpy() for k in range(len(nn))])
        out['rb_%s_pairs_out' % key] = np.concatenate([p[k, 1, :nn[k]].numpy() for k in range(len(nn))])
    x = torch.from_numpy(feats)
    acts = []
    for li, (key, subm, cin, cout, ks, st, pd) in enumerate(LAYERS):
        inp, o, p, n, ish, osh = books[key]
        w = torch.from_numpy(layer_weight(li, ks, cin, cout))
        y = ref.indice_conv(x, w, p, n, o.shape[0], False, subm)
        out['l%d_y' % li] = y.numpy()
        acts.append((x, w, p, n, subm))
        x = torch.relu(y)
    for li, (xi, w, p, n, subm) in enumerate(acts):
        dy = torch.from_numpy(layer_dy(li, out['l%d_y' % li].shape))
        dx, dw = ref.indice_conv_backward(xi, w, dy, p, n, False, subm)
        out['l%d_dx' % li] = dx.numpy()
        dw = dw.numpy().reshape(-1, w.shape[-2], w.shape[-1])
        if w.shape[-2] * w.shape[-1] >= 32 * 32 and dw.shape[0] == 27:
            out['l%d_dw_taps' % li] = np.array([0, 13, 26])
            dw = dw[[0, 13, 26]]
        out['l%d_dw' % li] = dw
    np.savez_compressed(os.path.join(HERE, 'spconv_ref_small.npz'), **out)
    print('small: %d voxels; rulebook outputs' % len(idx),
          {k: int(v[1].shape[0]) for k, v in books.items()})

    # ---------------- full frames: digests ----------------
    feats, idx = voxelize([f['points'] for f in frames])
    books = rulebooks(ref, idx, 2)
    dig = dict(seeds=np.array([0, 1]), n_voxels=np.array(len(idx)), indices_sha1=np.array(sha(idx)))
    for key, (inp, o, p, n, ish, osh) in books.items():
        nn = n.numpy()
        c_pairs, c_out = canonical(inp.numpy(), ish, o.numpy(), p.numpy(), nn, osh)
        dig['%s_num' % key] = nn
        dig['%s_n_out' % key] = np.array(o.shape[0])
        dig['%s_outids_sha1' % key] = np.array(sha(o.numpy()))
        dig['%s_pairs_sha1' % key] = np.array(sha(p.numpy()))
        dig['%s_canon_pairs_sha1' % key] = np.array(c_pairs)
        dig['%s_canon_out_sha1' % key] = np.array(c_out)
    np.savez_compressed(os.path.join(HERE, 'spconv_ref_full.npz'), **dig)
    print('full: %d voxels' % len(idx), {k: int(v[1].shape[0]) for k, v in books.items()})
    for f in ('spconv_ref_small.npz', 'spconv_ref_full.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
