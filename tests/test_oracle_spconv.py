"""Oracle pinning, sparse conv: SURVEY K2/K3 KATs (produced by the reference's compiled CPU
functors) plus an independent brute-force definition on random small inputs."""
import numpy as np
import pytest

IDX = np.array([[0, 5, 5, 5], [0, 5, 5, 6], [0, 6, 5, 5], [0, 9, 9, 9]], np.int32)
SHAPE = [41, 1600, 1408]
W = np.stack([(k + 1) * np.ones((4, 1), np.float32) for k in range(27)])


def test_k2_subm_kat(orc):
    o, p, n, _ = orc.get_indice_pairs(IDX, 1, SHAPE, [3, 3, 3], [1, 1, 1], [1, 1, 1], subm=True)
    assert {k: int(x) for k, x in enumerate(n) if x} == {4: 1, 5: 1, 12: 1, 13: 4, 14: 1, 21: 1,
                                                         22: 1}
    want = {4: (0, 2), 5: (1, 2), 12: (0, 1), 14: (1, 0), 21: (2, 1), 22: (2, 0)}
    for k, pr in want.items():
        assert tuple(p[k, :, 0]) == pr
    assert p[13, :, :4].tolist() == [[0, 1, 2, 3], [0, 1, 2, 3]]
    out = orc.indice_conv(np.eye(4, dtype=np.float32), W, p, n, 4, subm=True)
    assert out.ravel().tolist() == [52, 49, 25, 14]


def test_k2_strided_kat(orc):
    o, p, n, osh = orc.get_indice_pairs(IDX, 1, SHAPE, [3, 3, 3], [2, 2, 2], [1, 1, 1],
                                        subm=False, sort_out=False)
    assert osh == [21, 800, 704] and len(o) == 16
    assert o[:, 1:].tolist() == [[3, 3, 3], [3, 3, 2], [3, 2, 3], [3, 2, 2], [2, 3, 3], [2, 3, 2],
                                 [2, 2, 3], [2, 2, 2], [5, 5, 5], [5, 5, 4], [5, 4, 5], [5, 4, 4],
                                 [4, 5, 5], [4, 5, 4], [4, 4, 5], [4, 4, 4]]
    assert {k: int(x) for k, x in enumerate(n) if x} == {
        0: 2, 1: 1, 2: 2, 6: 2, 7: 1, 8: 2, 9: 1, 11: 1, 15: 1, 17: 1, 18: 2, 19: 1, 20: 2, 24: 2,
        25: 1, 26: 2}
    assert p[0, :, :2].T.tolist() == [[0, 0], [3, 8]]
    assert p[1, :, :1].T.tolist() == [[1, 0]] and p[9, :, :1].T.tolist() == [[2, 0]]
    out = orc.indice_conv(np.eye(4, dtype=np.float32), W, p, n, 16)
    assert out.ravel().tolist() == [13, 15, 31, 27, 39, 21, 51, 27, 1, 3, 7, 9, 19, 21, 25, 27]


def test_sorted_output_order(orc):
    o, p, n, osh = orc.get_indice_pairs(IDX, 1, SHAPE, [3, 3, 3], [2, 2, 2], [1, 1, 1],
                                        subm=False, sort_out=True)
    flat = (o[:, 1] * osh[1] + o[:, 2]) * osh[2] + o[:, 3]
    assert np.all(np.diff(flat) > 0)
    out = orc.indice_conv(np.eye(4, dtype=np.float32), W, p, n, 16)
    assert sorted(out.ravel().tolist()) == sorted(
        [13, 15, 31, 27, 39, 21, 51, 27, 1, 3, 7, 9, 19, 21, 25, 27])


def _brute(indices, feats, w, ksize, stride, pad, out_shape, subm):
    """Definition: out[q] = sum_k feat[p = q*s - pad + k] @ W[k]; outputs = sorted touched cells."""
    kz, ky, kx = ksize
    pos = {tuple(r): i for i, r in enumerate(indices.tolist())}
    if subm:
        outs = [tuple(r) for r in indices.tolist()]
    else:
        outs = set()
        for b, z, y, x in indices.tolist():
            for a in range(kz):
                for bb in range(ky):
                    for c in range(kx):
                        t = (z + pad[0] - a, y + pad[1] - bb, x + pad[2] - c)
                        if all(v >= 0 and v % s == 0 and v // s < o
                               for v, s, o in zip(t, stride, out_shape)):
                            outs.add((b,) + tuple(v // s for v, s in zip(t, stride)))
        outs = sorted(outs)
    res = np.zeros((len(outs), w.shape[-1]), np.float64)
    for r, (b, z, y, x) in enumerate(outs):
        for a in range(kz):
            for bb in range(ky):
                for c in range(kx):
                    p = (b, z * stride[0] - pad[0] + a, y * stride[1] - pad[1] + bb,
                         x * stride[2] - pad[2] + c)
                    if p in pos:
                        res[r] += feats[pos[p]].astype(np.float64) @ w[(a * ky + bb) * kx + c]
    return np.array(outs, np.int32).reshape(-1, 4), res


@pytest.mark.parametrize('cfg', [
    dict(ks=[3, 3, 3], st=[1, 1, 1], pd=[1, 1, 1], subm=True),
    dict(ks=[3, 3, 3], st=[2, 2, 2], pd=[1, 1, 1], subm=False),
    dict(ks=[3, 3, 3], st=[2, 2, 2], pd=[0, 1, 1], subm=False),
    dict(ks=[3, 1, 1], st=[2, 1, 1], pd=[0, 0, 0], subm=False),
])
def test_against_bruteforce(orc, cfg):
    rng = np.random.default_rng(7)
    shape = [9, 12, 10]
    cells = rng.choice(2 * 9 * 12 * 10, size=150, replace=False)
    b, rem = np.divmod(cells, 9 * 12 * 10)
    z, rem = np.divmod(rem, 120)
    y, x = np.divmod(rem, 10)
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    idx = idx[np.lexsort((x, y, z, b))]
    cin, cout = 5, 3
    kvol = int(np.prod(cfg['ks']))
    feats = rng.standard_normal((150, cin)).astype(np.float32)
    w = rng.standard_normal((kvol, cin, cout)).astype(np.float32)
    o, p, n, osh = orc.get_indice_pairs(idx, 2, shape, cfg['ks'], cfg['st'], cfg['pd'],
                                        subm=cfg['subm'], sort_out=True)
    out = orc.indice_conv(feats, w, p, n, len(o), subm=cfg['subm'])
    bo, bres = _brute(idx, feats, w, cfg['ks'], cfg['st'], cfg['pd'], osh, cfg['subm'])
    assert np.array_equal(o, bo)
    np.testing.assert_allclose(out, bres, rtol=1e-5, atol=1e-5)
    # backward == adjoint of forward (linearity): <dY, conv(X)> == <dX, X> and == <dW, W>
    dy = rng.standard_normal(out.shape).astype(np.float32)
    dx, dw = orc.indice_conv_backward(feats, w, dy, p, n, subm=cfg['subm'])
    lhs = float((dy.astype(np.float64) * bres).sum())
    np.testing.assert_allclose(float((dx.astype(np.float64) * feats).sum()), lhs, rtol=1e-4)
    np.testing.assert_allclose(float((dw.astype(np.float64) * w).sum()), lhs, rtol=1e-4)


def test_empty_input(orc):
    o, p, n, _ = orc.get_indice_pairs(np.zeros((0, 4), np.int32), 1, SHAPE, [3, 3, 3], [1, 1, 1],
                                      [1, 1, 1], subm=True)
    assert len(o) == 0 and int(n.sum()) == 0


def test_int32_limit(orc):
    with pytest.raises(ValueError):
        orc.get_indice_pairs(IDX, 24, SHAPE, [3, 3, 3], [1, 1, 1], [1, 1, 1], subm=True)


# ---------------------------------------------------------------------------------------------
# Reference-compiled goldens (tests/golden/gen_spconv_golden.py: the reference's unpatched
# geometry.h + reordering.cc built by oracle/build_ref.py).  These pin the oracle's rulebook bit
# for bit (first-touch order, raw indicePairs bytes) and its conv forward / backward to 1e-5.
import hashlib  # noqa: E402
import os  # noqa: E402

from conftest import GOLDEN  # noqa: E402

LAYERS = [
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


def _sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _oracle_books(orc, idx, batch, sort_out=False):
    books, shape, cur = {}, SHAPE, idx
    for key, subm, cin, cout, ks, st, pd in LAYERS:
        if key in books:
            continue
        o, p, n, osh = orc.get_indice_pairs(cur, batch, shape, ks, st, pd, subm=subm,
                                            sort_out=sort_out)
        if subm:
            o = cur
        books[key] = (o, p, n, osh)
        cur, shape = o, osh
    return books


def test_rulebook_equals_reference_small(orc):
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_small.npz'))
    books = _oracle_books(orc, g['indices'], 2)
    for key, (o, p, n, osh) in books.items():
        assert osh == g['rb_%s_out_shape' % key].tolist()
        assert np.array_equal(n, g['rb_%s_num' % key]), key
        assert np.array_equal(o, g['rb_%s_outids' % key]), key          # first-touch order
        pin = np.concatenate([p[k, 0, :n[k]] for k in range(len(n))])
        pout = np.concatenate([p[k, 1, :n[k]] for k in range(len(n))])
        assert np.array_equal(pin, g['rb_%s_pairs_in' % key]), key      # pair ORDER too
        assert np.array_equal(pout, g['rb_%s_pairs_out' % key]), key


def test_conv_chain_equals_reference_small(orc):
    """12-layer chain, forward + input gradient + weight gradient vs the reference's
    gather / torch::mm / scatter-add sequence: <= 1e-5 (MKL vs plain-C summation order)."""
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_small.npz'))
    books = _oracle_books(orc, g['indices'], 2)
    x = g['features']
    for li, (key, subm, cin, cout, ks, st, pd) in enumerate(LAYERS):
        o, p, n, osh = books[key]
        w = layer_weight(li, ks, cin, cout).reshape(-1, cin, cout)
        y = orc.indice_conv(x, w, p, n, len(o), subm=subm)
        np.testing.assert_allclose(y, g['l%d_y' % li], rtol=1e-5, atol=1e-5, err_msg='fwd %d' % li)
        dx, dw = orc.indice_conv_backward(x, w, layer_dy(li, y.shape), p, n, subm=subm)
        np.testing.assert_allclose(dx, g['l%d_dx' % li], rtol=1e-5, atol=1e-5, err_msg='dx %d' % li)
        if 'l%d_dw_taps' % li in g:
            dw = dw[g['l%d_dw_taps' % li]]
        np.testing.assert_allclose(dw, g['l%d_dw' % li], rtol=1e-5, atol=2e-5, err_msg='dw %d' % li)
        x = np.maximum(g['l%d_y' % li], 0)        # chain on the reference's activations


def test_rulebook_digests_full_frames(orc):
    """Full KITTI-shaped frames (B=2, 27 k voxels, all 8 rulebooks): the oracle's raw outids and
    indicePairs arrays hash to the reference's, byte for byte."""
    from detmatch_amd import synth
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_full.npz'))
    feats, coors = [], []
    for b, s in enumerate(g['seeds']):
        v, c, n = orc.hard_voxelize(synth.lidar_frame(int(s))['points'], synth.KITTI_VOXEL,
                                    synth.KITTI_RANGE, 5, 16000)
        coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
    idx = np.concatenate(coors).astype(np.int32)
    assert _sha(idx) == str(g['indices_sha1'])
    for key, (o, p, n, osh) in _oracle_books(orc, idx, 2).items():
        assert np.array_equal(n, g['%s_num' % key]), key
        assert len(o) == int(g['%s_n_out' % key])
        if not key.startswith('subm'):
            assert _sha(o) == str(g['%s_outids_sha1' % key]), key
        assert _sha(p) == str(g['%s_pairs_sha1' % key]), key


def test_canonical_digest_is_order_free(orc):
    """The order-free digests of the golden (what the HIP rulebook, which sorts strided outputs
    by cell id, is held to) are reproduced by the oracle in its SORTED output mode."""
    from detmatch_amd import synth
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_full.npz'))
    coors = []
    for b, s in enumerate(g['seeds']):
        v, c, n = orc.hard_voxelize(synth.lidar_frame(int(s))['points'], synth.KITTI_VOXEL,
                                    synth.KITTI_RANGE, 5, 16000)
        coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
    idx = np.concatenate(coors).astype(np.int32)

    def cells(ids, shape):
        ids = ids.astype(np.int64)
        return ((ids[:, 0] * shape[0] + ids[:, 1]) * shape[1] + ids[:, 2]) * shape[2] + ids[:, 3]

    in_ids, in_shape = idx, SHAPE
    for key, (o, p, n, osh) in _oracle_books(orc, idx, 2, sort_out=True).items():
        ic, oc = cells(in_ids, in_shape), cells(o, osh)
        assert _sha(np.sort(oc)) == str(g['%s_canon_out_sha1' % key])
        h = hashlib.sha1()
        for k in range(len(n)):
            i, oo = ic[p[k, 0, :n[k]]], oc[p[k, 1, :n[k]]]
            order = np.lexsort((oo, i))
            h.update(np.ascontiguousarray(np.stack([i[order], oo[order]], 1)).tobytes())
        assert h.hexdigest() == str(g['%s_canon_pairs_sha1' % key]), key
        in_ids, in_shape = o, osh
