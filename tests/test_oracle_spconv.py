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
