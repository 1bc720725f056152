"""HIP sparse conv (rulebook + fused gather-GEMM + backward) vs the oracle, through the
C-ABI.  Integer outputs bit-exact (output voxel list, pair SETS — the reference's GPU slot
order is an atomicAdd race, so order inside a pair list is not part of the contract);
fp32 features within rtol 1e-4 / atol 1e-5 (summation order differs: the oracle adds
offset by offset like the reference, the kernel accumulates fma chains in MFMA order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPE = [41, 1600, 1408]
IDX = np.array([[0, 5, 5, 5], [0, 5, 5, 6], [0, 6, 5, 5], [0, 9, 9, 9]], np.int32)
RTOL, ATOL = 1e-4, 1e-5


def _pair_set(pairs, num, k):
    return set(map(tuple, pairs[k, :, :num[k]].T.tolist()))


def _build(dev, idx, batch, shape, ks, st, pd, subm):
    from detmatch_amd.spconv import ops
    t = torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    return ops.build_rulebook(t, batch, shape, ks, st, pd, 1, subm)


def _check_rulebook(orc, dev, idx, batch, shape, ks, st, pd, subm):
    rb = _build(dev, idx, batch, shape, ks, st, pd, subm)
    o, p, n, osh = orc.get_indice_pairs(idx, batch, shape, ks, st, pd, subm=subm, sort_out=True)
    assert rb.out_shape == osh
    assert np.array_equal(rb.outids.cpu().numpy(), o)            # output voxels + order: bit-exact
    gn = rb.indice_num.cpu().numpy()
    assert np.array_equal(gn, n)
    gp = rb.indice_pairs.cpu().numpy()
    kvol = len(n)
    for k in range(kvol):
        assert _pair_set(gp, gn, k) == _pair_set(p, n, k)
        assert np.all(gp[k, :, gn[k]:] == -1)
    # gather tables are consistent with the pair lists
    no = rb.nbr_out.cpu().numpy()
    for k in range(kvol):
        rows = np.nonzero(no[k] >= 0)[0]
        assert set(zip(no[k][rows].tolist(), rows.tolist())) == _pair_set(p, n, k)
    if not subm:
        ni = rb.nbr_in.cpu().numpy()
        for k in range(kvol):
            rows = np.nonzero(ni[k] >= 0)[0]
            assert set(zip(rows.tolist(), ni[k][rows].tolist())) == _pair_set(p, n, k)
    return rb, (o, p, n)


def test_survey_k2_kats(orc, dev):
    from detmatch_amd.spconv import ops
    W = np.stack([(k + 1) * np.ones((4, 16), np.float32) for k in range(27)])
    rb, _ = _check_rulebook(orc, dev, IDX, 1, SHAPE, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    feats = torch.eye(4, device=dev)
    out = ops.indice_conv(feats, torch.from_numpy(W).to(dev), rb.indice_pairs, rb.indice_num, 4,
                          False, True)
    assert out[:, 0].cpu().tolist() == [52, 49, 25, 14]
    rb, (o, p, n) = _check_rulebook(orc, dev, IDX, 1, SHAPE, [3, 3, 3], [2, 2, 2], [1, 1, 1],
                                    False)
    out = ops.indice_conv(feats, torch.from_numpy(W).to(dev), rb.indice_pairs, rb.indice_num, 16)
    want = orc.indice_conv(np.eye(4, dtype=np.float32), W, p, n, 16)
    assert np.array_equal(out.cpu().numpy(), want)   # small integers: exact
    assert sorted(want[:, 0].tolist()) == sorted(
        [13, 15, 31, 27, 39, 21, 51, 27, 1, 3, 7, 9, 19, 21, 25, 27])


def _rand_indices(rng, n, batch, shape):
    vol = shape[0] * shape[1] * shape[2]
    cells = np.sort(rng.choice(batch * vol, size=n, replace=False))
    b, rem = np.divmod(cells, vol)
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


@pytest.mark.parametrize('cfg', [
    dict(ks=[3, 3, 3], st=[1, 1, 1], pd=[1, 1, 1], subm=True),
    dict(ks=[3, 3, 3], st=[2, 2, 2], pd=[1, 1, 1], subm=False),
    dict(ks=[3, 3, 3], st=[2, 2, 2], pd=[0, 1, 1], subm=False),
    dict(ks=[3, 1, 1], st=[2, 1, 1], pd=[0, 0, 0], subm=False),
])
@pytest.mark.parametrize('n', [1, 37, 700])
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_rulebook_random(orc, dev, cfg, n, mode):
    """mode 0: automatic choice, 1: hash + radix sort, 2: occupancy bitmap (strided convs only)."""
    from detmatch_amd import _lib
    if mode and cfg['subm']:
        pytest.skip('sub-manifold rulebooks have one implementation')
    _lib.check(_lib.lib().dm_rulebook_set_mode(mode), 'dm_rulebook_set_mode')
    try:
        _rulebook_random(orc, dev, cfg, n)
    except RuntimeError as e:
        if mode == 2 and 'workspace' in str(e).lower():
            pytest.skip('bitmap larger than the workspace of %d rows' % n)
        raise
    finally:
        _lib.lib().dm_rulebook_set_mode(0)


def test_rulebook_bitmap_equals_hash_path_full_frame(dev):
    """The two strided-rulebook implementations on the four strided layers of a KITTI-sized frame
    pair: every output (cells, both tables, pair lists in slot order, counts) identical."""
    from detmatch_amd import _lib, synth, voxel
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    from detmatch_amd.spconv import ops
    pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
    _, coors, _, _, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    idx, shape, seen, n_strided = coors, [41, 1600, 1408], set(), 0
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key in seen:
            continue
        seen.add(key)
        books = []
        for mode in ((1, 2) if not subm else (0,)):
            _lib.check(_lib.lib().dm_rulebook_set_mode(mode), 'dm_rulebook_set_mode')
            try:
                books.append(ops.build_rulebook(idx, 2, shape, ks, st, pd, 1, subm))
            finally:
                _lib.lib().dm_rulebook_set_mode(0)
        if not subm:
            a, b = books
            n_strided += 1
            assert a.n_out == b.n_out and a.out_shape == b.out_shape
            for f in ('outids', 'nbr_out', 'nbr_in', 'indice_pairs', 'indice_num'):
                assert torch.equal(getattr(a, f), getattr(b, f)), (key, f)
        idx, shape = books[-1].outids, books[-1].out_shape
    assert n_strided == 4


def test_deferred_rulebooks_equal_two_phase(dev):
    """voxelize -> subm1 -> spconv2 -> ... -> spconv_down2 of a KITTI-sized frame pair issued at CAPACITY
    with every count on the device (dm_rulebook_subm_cap / dm_rulebook_conv_cap), counts read ONCE at the end: every
    table, pair list and count identical to the two-phase builds (one read-back per strided level)."""
    from detmatch_amd import synth, voxel
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    from detmatch_amd.spconv import ops
    pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
    v, coors, n, mean, counts = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000, sync=False)
    assert coors.shape[0] == 32000
    # capacity chain, no host value in between
    idx, n_dev, cap, shape, seen, caps = coors, counts[2:3], 32000, [41, 1600, 1408], {}, []
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key in seen:
            continue
        cap_out = None if subm else ops.strided_capacity(cap, ks, st, 32000)
        c = ops.build_rulebook_cap(idx, n_dev, cap, 2, shape, ks, st, pd, 1, subm, cap_out=cap_out)
        assert c is not None, key
        seen[key] = c
        caps.append((key, c, subm))
        idx, n_dev, cap, shape = c.outids, c.n_out_dev, c.cap_out, c.out_shape
    vals = torch.cat([counts[2:3]] + [c.n_out_dev for _, c, subm in caps if not subm]).tolist()      # THE read-back
    assert len(vals) == 5
    # two-phase reference
    idx, shape, n_in, k = coors[:vals[0]], [41, 1600, 1408], vals[0], 1
    for key, c, subm in caps:
        want = ops.build_rulebook(idx, 2, shape, *[(ks, st, pd) for kk, sm, _, _, ks, st, pd in BACKBONE_LAYERS if kk == key][0],
                                  1, subm)
        n_out = n_in if subm else vals[k]
        k += 0 if subm else 1
        got = c.finish(n_in, n_out)
        assert got is not None and got.n_out == want.n_out == n_out and got.out_shape == want.out_shape
        for f in ('outids', 'nbr_out', 'indice_pairs', 'indice_num') + (() if subm else ('nbr_in',)):
            a, b = getattr(got, f), getattr(want, f)
            assert a.shape == b.shape and torch.equal(a, b), (key, f)
        idx, shape, n_in = want.outids, want.out_shape, n_out
    # a capacity that is too small is reported, not silently truncated
    key, c, _ = [t for t in caps if not t[2]][0]
    small = ops.build_rulebook_cap(coors, counts[2:3], 32000, 2, [41, 1600, 1408], [3, 3, 3], [2, 2, 2], [1, 1, 1], 1, False,
                                   cap_out=1024)
    assert small.finish(vals[0], int(small.n_out_dev.item())) is None


def test_geometry_of_a_pass_needs_one_read_back(dev, monkeypatch):
    """OpenPCDetDetector.prepare_geometry_steps: with the deferred builds the generator asks ONCE (voxel count + the four
    N_out in one tensor); the batch it prepares equals the stepwise one (five asks)."""
    from detmatch_amd.mm3d import openpcdet
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, dev)
    det = wl.model.student.detector_3d
    from detmatch_amd import synth
    pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
    out = []
    for defer in (True, False):
        monkeypatch.setattr(openpcdet, 'DEFER_GEOMETRY_READBACKS', defer)
        det._geom_cache.clear()
        gen, asks = det.prepare_geometry_steps(pts, None), 0
        try:
            ask = next(gen)
            while True:
                asks += 1
                v = ask.reshape(-1).tolist()
                ask = gen.send(v[0] if len(v) == 1 else v)
        except StopIteration:
            pass
        out.append((asks, det._geom_cache[id(pts)][2]))
    (a1, r1), (a2, r2) = out
    assert a1 == 1 and a2 == 5
    assert torch.equal(r1['voxel_coords'], r2['voxel_coords']) and torch.equal(r1['voxel_features'], r2['voxel_features'])
    assert set(r1['indice_dict_prefetch']) == set(r2['indice_dict_prefetch'])
    for key in r1['indice_dict_prefetch']:
        (o1, i1, p1, n1, s1), (o2, i2, p2, n2, s2) = r1['indice_dict_prefetch'][key], r2['indice_dict_prefetch'][key]
        assert torch.equal(o1, o2) and torch.equal(i1, i2) and torch.equal(p1, p2) and torch.equal(n1, n2) and s1 == s2
        for t1, t2 in zip(p1.dm_tables[:2], p2.dm_tables[:2]):
            assert (t1 is None and t2 is None) or torch.equal(t1, t2)


@pytest.mark.parametrize('profile', ['kitti', 'waymo'])
def test_strided_rulebook_properties_full_size(dev, profile):
    """Size-independent properties of the strided rulebooks at BASELINE's full frame sizes (KITTI B = 2, the
    Waymo-shaped 200 k-point sweep at B = 1): output cells strictly ascending by flat cell id (the reference's
    torch::_unique order) and inside the output grid; every output reached by at least one pair; nbr_in and nbr_out
    are each other's transpose; the pair lists hold exactly the table's pairs, padded with -1; indice_num counts
    them; every pair obeys out = (in + pad - k) / stride."""
    from detmatch_amd import synth, voxel
    from detmatch_amd.spconv import ops
    if profile == 'kitti':
        pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
        vs, rng, shape, mv, batch = synth.KITTI_VOXEL, synth.KITTI_RANGE, [41, 1600, 1408], 16000, 2
    else:
        pts = [torch.from_numpy(synth.lidar_frame(7, full360=True)['points']).to(dev)]
        vs, rng, shape, mv, batch = synth.WAYMO_VOXEL, synth.WAYMO_RANGE, [41, 1504, 1504], 150000, 1      # grid (z, y, x)
    _, coors, _, _, _ = voxel.voxelize_batch(pts, vs, rng, 5, mv)
    idx = coors
    for level in range(3):
        ks, st, pd = [3, 3, 3], [2, 2, 2], ([1, 1, 1] if level < 2 else [0, 1, 1])
        rb = ops.build_rulebook(idx, batch, shape, ks, st, pd, 1, False)
        out, osh = rb.outids.long(), rb.out_shape
        flat = ((out[:, 0] * osh[0] + out[:, 1]) * osh[1] + out[:, 2]) * osh[2] + out[:, 3]
        assert bool((flat[1:] > flat[:-1]).all())
        for a in range(3):
            assert int(out[:, a + 1].min()) >= 0 and int(out[:, a + 1].max()) < osh[a]
        n_in, n_out = idx.shape[0], out.shape[0]
        assert bool((rb.nbr_out >= 0).any(dim=0).all())
        total = 0
        for k in range(27):
            ni, no = rb.nbr_in[k].long(), rb.nbr_out[k].long()
            src = torch.nonzero(ni >= 0).flatten()
            dst = ni[src]
            assert torch.equal(no[dst], src)
            assert int((no >= 0).sum()) == src.numel() == int(rb.indice_num[k])
            m = src.numel()
            pin, pout = rb.indice_pairs[k, 0].long(), rb.indice_pairs[k, 1].long()
            assert bool((pin[m:] == -1).all()) and bool((pout[m:] == -1).all())
            assert torch.equal(ni[pin[:m]], pout[:m]) and pin[:m].unique().numel() == m
            kk = (k // 9, (k // 3) % 3, k % 3)
            a_in, a_out = idx.long()[src], out[dst]
            assert torch.equal(a_in[:, 0], a_out[:, 0])
            for a in range(3):
                assert torch.equal(a_in[:, a + 1] + pd[a] - kk[a], a_out[:, a + 1] * st[a])
            total += m
        assert total >= n_in          # every input reaches at least one output
        idx, shape = rb.outids, osh


def _rulebook_random(orc, dev, cfg, n):
    rng = np.random.default_rng(n)
    shape = [9, 14, 11]
    idx = _rand_indices(rng, n, 3, shape)
    idx = idx[rng.permutation(n)]   # arbitrary (non-sorted) input order, like voxel order
    # keep each sample's rows contiguous as the stacked layout requires
    idx = idx[np.argsort(idx[:, 0], kind='stable')]
    _check_rulebook(orc, dev, idx, 3, shape, cfg['ks'], cfg['st'], cfg['pd'], cfg['subm'])


@pytest.mark.parametrize('cin,cout', [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64),
                                      (64, 128)])
@pytest.mark.parametrize('subm', [True, False])
def test_conv_forward_backward(orc, dev, cin, cout, subm):
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(cin * 1000 + cout)
    shape = [12, 40, 40]
    n = 3000
    idx = _rand_indices(rng, n, 2, shape)
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    rb, (o, p, nn) = _check_rulebook(orc, dev, idx, 2, shape, ks, st, pd, subm)
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    tf = torch.from_numpy(feats).to(dev)
    tw = torch.from_numpy(w).to(dev).view(3, 3, 3, cin, cout)
    out = ops.indice_conv(tf, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    want = orc.indice_conv(feats, w, p, nn, len(o), subm=subm)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    dy = rng.standard_normal(want.shape).astype(np.float32)
    need_dx = cin >= 16
    dx, dw = ops.indice_conv_backward(tf, tw, torch.from_numpy(dy).to(dev), rb.indice_pairs,
                                      rb.indice_num, False, subm, need_input_grad=need_dx)
    odx, odw = orc.indice_conv_backward(feats, w, dy, p, nn, subm=subm)
    np.testing.assert_allclose(dw.cpu().numpy().reshape(27, cin, cout), odw, rtol=1e-3, atol=1e-3)
    if need_dx:
        np.testing.assert_allclose(dx.cpu().numpy(), odx, rtol=RTOL, atol=ATOL)


def test_foreign_rulebook_compat_path(orc, dev):
    """indice_conv on reference-format pair lists that did NOT come from our builder."""
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(3)
    shape = [12, 40, 40]
    idx = _rand_indices(rng, 2000, 1, shape)
    o, p, nn, _ = orc.get_indice_pairs(idx, 1, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], subm=False)
    feats = rng.standard_normal((2000, 16)).astype(np.float32)
    w = (rng.standard_normal((27, 16, 32)) * 0.1).astype(np.float32)
    tp = torch.from_numpy(p).to(dev)
    tn = torch.from_numpy(nn).to(dev)
    out = ops.indice_conv(torch.from_numpy(feats).to(dev), torch.from_numpy(w).to(dev), tp, tn,
                          len(o))
    np.testing.assert_allclose(out.cpu().numpy(), orc.indice_conv(feats, w, p, nn, len(o)),
                               rtol=RTOL, atol=ATOL)
    dy = rng.standard_normal((len(o), 32)).astype(np.float32)
    dx, dw = ops.indice_conv_backward(torch.from_numpy(feats).to(dev), torch.from_numpy(w).to(dev),
                                      torch.from_numpy(dy).to(dev), tp, tn)
    odx, odw = orc.indice_conv_backward(feats, w, dy, p, nn)
    np.testing.assert_allclose(dx.cpu().numpy(), odx, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(dw.cpu().numpy(), odw, rtol=1e-3, atol=1e-3)


def test_empty_input(dev):
    from detmatch_amd.spconv import ops
    idx = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    rb = ops.build_rulebook(idx, 1, SHAPE, 3, 1, 1, 1, True)
    assert rb.n_out == 0 and int(rb.indice_num.sum().item()) == 0
    rb = ops.build_rulebook(idx, 1, SHAPE, 3, 2, 1, 1, False)
    assert rb.n_out == 0 and rb.outids.shape == (0, 4)
    out = ops.indice_conv(torch.zeros((0, 16), device=dev), torch.zeros((3, 3, 3, 16, 32), device=dev),
                          rb.indice_pairs, rb.indice_num, 0)
    assert out.shape == (0, 32)


def test_int32_limit_is_an_error(dev):
    from detmatch_amd import _lib
    from detmatch_amd.spconv import ops
    with pytest.raises(_lib.DetMatchHipError):
        ops.build_rulebook(torch.from_numpy(IDX).to(dev), 24, SHAPE, 3, 1, 1, 1, True)


def test_backbone_chain_kitti_shape(orc, dev):
    """The 12 sparse convs of VoxelBackBone8x (spconv_backbone.py:80-120) chained on two
    KITTI-shaped frames, module API + autograd, against the oracle layer by layer."""
    from detmatch_amd import synth, voxel
    from detmatch_amd import spconv
    frames = [synth.lidar_frame(s)['points'] for s in (0, 1)]
    t = [torch.from_numpy(p).to(dev) for p in frames]
    _, coors, _, mean, _ = voxel.voxelize_batch(t, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    layers = [('subm1', True, 4, 16, 3, 1, 1), ('subm1', True, 16, 16, 3, 1, 1),
              ('spconv2', False, 16, 32, 3, 2, 1), ('subm2', True, 32, 32, 3, 1, 1),
              ('subm2', True, 32, 32, 3, 1, 1), ('spconv3', False, 32, 64, 3, 2, 1),
              ('subm3', True, 64, 64, 3, 1, 1), ('subm3', True, 64, 64, 3, 1, 1),
              ('spconv4', False, 64, 64, 3, 2, (0, 1, 1)), ('subm4', True, 64, 64, 3, 1, 1),
              ('subm4', True, 64, 64, 3, 1, 1),
              ('spconv_down2', False, 64, 128, (3, 1, 1), (2, 1, 1), 0)]
    torch.manual_seed(0)
    x = spconv.SparseConvTensor(mean.clone().requires_grad_(False), coors, SHAPE, 2)
    ox_idx, ox_feat, oshape = coors.cpu().numpy(), mean.cpu().numpy(), SHAPE
    books = {}
    total_pairs = 0
    for key, subm, cin, cout, ks, st, pd in layers:
        cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
        m = cls(cin, cout, ks, stride=st, padding=pd, bias=False, indice_key=key).to(dev)
        with torch.no_grad():
            m.weight.mul_(3.0)   # keep activations O(1) through 12 layers
        y = m(x)
        as3 = lambda v: list(v) if isinstance(v, (tuple, list)) else [v] * 3
        if key not in books:
            books[key] = orc.get_indice_pairs(ox_idx, 2, oshape, as3(ks), as3(st), as3(pd),
                                              subm=subm, sort_out=True)
        o, p, n, osh = books[key]
        total_pairs += int(n.sum())
        w = m.weight.detach().cpu().numpy().reshape(-1, cin, cout)
        want = orc.indice_conv(ox_feat, w, p, n, len(o), subm=subm)
        assert np.array_equal(y.indices.cpu().numpy(), o)
        got = y.features.detach().cpu().numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL * scale)
        # next layer input: ReLU of the ORACLE output on both sides (no drift accumulation)
        ox_idx, oshape = o, osh
        ox_feat = np.maximum(want, 0)
        x = spconv.SparseConvTensor(torch.from_numpy(ox_feat).to(dev), y.indices, osh, 2)
        x.indice_dict = y.indice_dict
    assert total_pairs > 1_000_000   # realistic sub-manifold neighbour counts


def test_autograd_through_modules(orc, dev):
    from detmatch_amd import spconv
    rng = np.random.default_rng(11)
    shape = [12, 40, 40]
    idx = _rand_indices(rng, 2500, 2, shape)
    feats = rng.standard_normal((2500, 16)).astype(np.float32)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key='s1'),
        torch.nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='c2'),
    ).to(dev)
    tf = torch.from_numpy(feats).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(tf, torch.from_numpy(idx).to(dev), shape, 2)
    y = net(x)
    dy = rng.standard_normal(tuple(y.features.shape)).astype(np.float32)
    y.features.backward(torch.from_numpy(dy).to(dev))
    # oracle chain
    w1 = net[0].weight.detach().cpu().numpy().reshape(27, 16, 16)
    w2 = net[2].weight.detach().cpu().numpy().reshape(27, 16, 32)
    o1, p1, n1, _ = orc.get_indice_pairs(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, subm=True)
    h = orc.indice_conv(feats, w1, p1, n1, len(o1), subm=True)
    hr = np.maximum(h, 0)
    o2, p2, n2, _ = orc.get_indice_pairs(o1, 2, shape, [3] * 3, [2] * 3, [1] * 3, subm=False)
    dh, dw2 = orc.indice_conv_backward(hr, w2, dy, p2, n2)
    dh = dh * (h > 0)
    dx, dw1 = orc.indice_conv_backward(feats, w1, dh, p1, n1, subm=True)
    np.testing.assert_allclose(tf.grad.cpu().numpy(), dx, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(net[0].weight.grad.cpu().numpy().reshape(27, 16, 16), dw1,
                               rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(net[2].weight.grad.cpu().numpy().reshape(27, 16, 32), dw2,
                               rtol=1e-3, atol=1e-3)


def test_batched_weight_gradients_equal_per_layer_calls(dev, monkeypatch):
    """The weight gradients of a backward pass delivered by ONE dm_spconv_wgrad_batch call at the end of the pass
    (queued by the autograd Functions inside `deferred_weight_grads()`, flushed when the context is left) are bit-identical to the
    per-layer dm_spconv_wgrad results — first pass (gradients absent), second pass (accumulated in place), the
    4-channel input layer (rides on the 16-row tile) and a layer watched by a post-accumulate hook (never deferred:
    the hook must find the gradient) included."""
    from detmatch_amd import spconv
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(5)
    shape = [12, 48, 48]
    idx = _rand_indices(rng, 6000, 2, shape)
    feats = torch.from_numpy(rng.standard_normal((6000, 4)).astype(np.float32)).to(dev)
    torch.manual_seed(3)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(4, 16, 3, padding=1, bias=False, indice_key='s1'), torch.nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key='s1'), torch.nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='c2'), torch.nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key='s2'), torch.nn.ReLU(),
        spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False, indice_key='c3'), torch.nn.ReLU(),
        spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key='s3'),
        spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=0, bias=False, indice_key='d'),
    ).to(dev)
    fired = []
    net[4].weight.register_post_accumulate_grad_hook(lambda p: fired.append(1))
    res = {}
    for batch in (False, True):
        monkeypatch.setattr(ops, 'WGRAD_BATCH', batch)
        for p in net.parameters():
            p.grad = None
        before = ops.WGRAD_BATCHES[0]
        snaps = []
        for rep in range(2):          # the second pass accumulates into the first one's gradients
            x = spconv.SparseConvTensor(feats * (rep + 1), torch.from_numpy(idx).to(dev), shape, 2)
            y = net(x).features
            with ops.deferred_weight_grads():
                (y * y).sum().backward()
            snaps.append([p.grad.clone() for p in net.parameters()])
        res[batch] = snaps
        assert ops.WGRAD_BATCHES[0] - before == (2 if batch else 0)
    for sa, sb in zip(res[False], res[True]):
        for i, (a, b) in enumerate(zip(sa, sb)):
            if i == 0:      # 4 -> 16: the per-layer call uses another kernel (one wave per 16-channel block, other chunks)
                torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-6 * float(a.abs().max()))
            else:
                assert torch.equal(a, b), i
    assert len(fired) == 4


@pytest.mark.parametrize('subm', [True, False])
def test_tile_launch_order(orc, dev, subm, monkeypatch):
    """dm_spconv_tile_order: a permutation of the 16-row tiles by descending number of active kernel
    offsets; the gather-GEMM result does not depend on it (bit-exact with and without) and still
    matches the oracle."""
    from detmatch_amd.spconv import ops
    rng = np.random.default_rng(5)
    shape = [12, 60, 60]
    n = 9000
    idx = _rand_indices(rng, n, 2, shape)
    # plus a dense clump in sample 0 so that tiles differ in work
    clump = np.stack([np.zeros(4000, np.int64), rng.integers(2, 8, 4000), rng.integers(10, 30, 4000),
                      rng.integers(10, 30, 4000)], 1).astype(idx.dtype)
    idx = np.unique(np.concatenate([idx, clump]), axis=0)        # sorted: sample rows stay contiguous
    n = len(idx)
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    rb, (o, p, nn) = _check_rulebook(orc, dev, idx, 2, shape, ks, st, pd, subm)
    tables = [rb.nbr_out] + ([] if subm else [rb.nbr_in])
    tile = 16                                          # rows per tile of the default kernel (spconv_gr)
    for nbr in tables:
        order = ops.tile_order(nbr)
        rows = nbr.shape[1]
        nt = (rows + tile - 1) // tile
        order = order[:nt]
        assert torch.equal(torch.sort(order.long())[0], torch.arange(nt, device=dev))
        act = torch.zeros(nt * tile, nbr.shape[0], dtype=torch.bool, device=dev)
        act[:rows] = (nbr >= 0).t()
        work = act.view(nt, tile, -1).any(dim=1).sum(dim=1)
        w_sorted = work[order.long()]
        assert bool((w_sorted[:-1] >= w_sorted[1:]).all()) and int(work.max()) > int(work.min())
        assert ops.tile_order(nbr).data_ptr() == order.data_ptr()   # cached on the table
    cin = cout = 64
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((27, cin, cout)) * 0.1).astype(np.float32)
    tf = torch.from_numpy(feats).to(dev)
    tw = torch.from_numpy(w).to(dev).view(3, 3, 3, cin, cout)
    dy = torch.randn(rb.n_out, cout, device=dev)
    assert min(n, rb.n_out) >= ops.TILE_ORDER_MIN_ROWS
    monkeypatch.setattr(ops, 'PACK_ROWS', False)                 # the launch order alone: same bits
    out = ops.indice_conv(tf, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    dx, _ = ops.indice_conv_backward(tf, tw, dy, rb.indice_pairs, rb.indice_num, False, subm)
    monkeypatch.setattr(ops, 'TILE_ORDER_MIN_ROWS', 1 << 30)      # identity order
    out0 = ops.indice_conv(tf, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    dx0, _ = ops.indice_conv_backward(tf, tw, dy, rb.indice_pairs, rb.indice_num, False, subm)
    assert torch.equal(out, out0) and torch.equal(dx, dx0)
    want = orc.indice_conv(feats, w, p, nn, len(o), subm=subm)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    # rows packed by neighbour mask (dm_spconv_pack_rows): a permutation of the rows, masks ascending,
    # the table gathered accordingly; results equal up to the fp32 summation order over the offsets
    # (the offsets of a tile are dealt to its four waves by rank), run-to-run bitwise reproducible
    monkeypatch.setattr(ops, 'TILE_ORDER_MIN_ROWS', 4096)
    monkeypatch.setattr(ops, 'PACK_ROWS', True)
    for nbr in tables:
        packed, perm, order = ops.packed_rows(nbr)
        nt16 = (nbr.shape[1] + 15) // 16
        assert torch.equal(torch.sort(order[:nt16].long())[0], torch.arange(nt16, device=dev))
        rows = nbr.shape[1]
        assert torch.equal(torch.sort(perm.long())[0], torch.arange(rows, device=dev))
        assert torch.equal(packed, nbr[:, perm.long()])
        bits = (1 << torch.arange(nbr.shape[0], device=dev, dtype=torch.int64))[:, None]
        key = ((packed >= 0).long() * bits).sum(0)
        assert bool((key[:-1] <= key[1:]).all())
        same = key[:-1] == key[1:]
        assert bool((perm[:-1][same] < perm[1:][same]).all())      # stable: raster order inside a group
        assert ops.packed_rows(nbr)[0] is packed
    out1 = ops.indice_conv(tf, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    dx1, _ = ops.indice_conv_backward(tf, tw, dy, rb.indice_pairs, rb.indice_num, False, subm)
    out2 = ops.indice_conv(tf, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
    assert torch.equal(out1, out2)
    np.testing.assert_allclose(out1.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out1.cpu().numpy(), out0.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dx1.cpu().numpy(), dx0.cpu().numpy(), rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------
# HIP vs goldens produced by the REFERENCE sparse-conv CPU code compiled here
# (tests/golden/gen_spconv_golden.py).  The reference lists strided outputs in first-touch order,
# the HIP rulebook sorted by cell id (documented in DESIGN §2), so rows are matched through their
# voxel coordinates: the pair SETS (input row, output cell) and the output cell set are bit-exact.
import hashlib  # noqa: E402
import os  # noqa: E402

from conftest import GOLDEN  # noqa: E402
from test_oracle_spconv import LAYERS, layer_dy, layer_weight  # noqa: E402


def _cells(ids, shape):
    ids = ids.astype(np.int64)
    return ((ids[:, 0] * shape[0] + ids[:, 1]) * shape[1] + ids[:, 2]) * shape[2] + ids[:, 3]


def _hip_books(dev, idx, batch):
    from detmatch_amd.spconv import ops
    books, shape, cur = {}, SHAPE, torch.from_numpy(np.ascontiguousarray(idx)).to(dev)
    for key, subm, cin, cout, ks, st, pd in LAYERS:
        if key in books:
            continue
        rb = ops.build_rulebook(cur, batch, shape, ks, st, pd, 1, subm)
        books[key] = (rb, shape)
        cur, shape = rb.outids, rb.out_shape
    return books


def test_rulebook_and_conv_equal_reference_small(dev):
    from detmatch_amd.spconv import ops
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_small.npz'))
    books = _hip_books(dev, g['indices'], 2)
    perm = {}            # key -> for every reference output row, the HIP row holding the same voxel
    prev_perm = np.arange(len(g['indices']))
    for key, (rb, in_shape) in books.items():
        osh = g['rb_%s_out_shape' % key].tolist()
        assert rb.out_shape == osh
        ref_cells = _cells(g['rb_%s_outids' % key], osh)
        hip_cells = _cells(rb.outids.cpu().numpy(), osh)
        assert np.array_equal(np.sort(ref_cells), np.sort(hip_cells)), key       # same voxel set
        order = np.argsort(hip_cells)
        perm[key] = order[np.searchsorted(hip_cells[order], ref_cells)]
        num = g['rb_%s_num' % key]
        assert np.array_equal(rb.indice_num.cpu().numpy(), num), key
        gp = rb.indice_pairs.cpu().numpy()
        s = 0
        for k in range(len(num)):
            ref = set(zip(prev_perm[g['rb_%s_pairs_in' % key][s:s + num[k]]].tolist(),
                          perm[key][g['rb_%s_pairs_out' % key][s:s + num[k]]].tolist()))
            assert _pair_set(gp, num, k) == ref, (key, k)
            s += num[k]
        prev_perm = perm[key]
    # conv forward / backward of the 12 layers on the reference's activations
    x_ref, in_perm = g['features'], np.arange(len(g['indices']))
    for li, (key, subm, cin, cout, ks, st, pd) in enumerate(LAYERS):
        rb, _ = books[key]
        out_perm = perm[key]
        x = np.empty_like(x_ref)
        x[in_perm] = x_ref                      # reference row r lives in HIP row in_perm[r]
        w = layer_weight(li, ks, cin, cout)
        tx, tw = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
        y = ops.indice_conv(tx, tw, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        np.testing.assert_allclose(y.cpu().numpy()[out_perm], g['l%d_y' % li], rtol=1e-5, atol=1e-5,
                                   err_msg='fwd %d' % li)
        dy_ref = layer_dy(li, g['l%d_y' % li].shape)
        dy = np.empty_like(dy_ref)
        dy[out_perm] = dy_ref
        dx, dw = ops.indice_conv_backward(tx, tw, torch.from_numpy(dy).to(dev), rb.indice_pairs,
                                          rb.indice_num, False, subm, need_input_grad=cin >= 16)
        if cin >= 16:
            np.testing.assert_allclose(dx.cpu().numpy()[in_perm], g['l%d_dx' % li], rtol=1e-5,
                                       atol=1e-5, err_msg='dx %d' % li)
        dw = dw.cpu().numpy().reshape(-1, cin, cout)
        if 'l%d_dw_taps' % li in g:
            dw = dw[g['l%d_dw_taps' % li]]
        np.testing.assert_allclose(dw, g['l%d_dw' % li], rtol=1e-5, atol=2e-5, err_msg='dw %d' % li)
        x_ref, in_perm = np.maximum(g['l%d_y' % li], 0), out_perm


def test_rulebook_digests_full_frames_reference(dev):
    """Full KITTI-shaped frames (B=2, 27 k voxels): every rulebook's canonical (order-free) form
    hashes to the reference's."""
    from detmatch_amd import synth, voxel
    g = np.load(os.path.join(GOLDEN, 'spconv_ref_full.npz'))
    t = [torch.from_numpy(synth.lidar_frame(int(s))['points']).to(dev) for s in g['seeds']]
    _, coors, _, _, _ = voxel.voxelize_batch(t, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    idx = coors.cpu().numpy()
    assert hashlib.sha1(np.ascontiguousarray(idx).tobytes()).hexdigest() == str(g['indices_sha1'])
    in_ids = idx
    for key, (rb, in_shape) in _hip_books(dev, idx, 2).items():
        in_cells = _cells(in_ids, in_shape)
        num = rb.indice_num.cpu().numpy()
        assert np.array_equal(num, g['%s_num' % key]), key
        assert rb.n_out == int(g['%s_n_out' % key])
        cells = _cells(rb.outids.cpu().numpy(), rb.out_shape)
        assert hashlib.sha1(np.sort(cells).tobytes()).hexdigest() == str(g['%s_canon_out_sha1' % key])
        p = rb.indice_pairs.cpu().numpy()
        h = hashlib.sha1()
        for k in range(len(num)):
            i = in_cells[p[k, 0, :num[k]]]
            o = cells[p[k, 1, :num[k]]]
            order = np.lexsort((o, i))
            h.update(np.ascontiguousarray(np.stack([i[order], o[order]], 1)).tobytes())
        assert h.hexdigest() == str(g['%s_canon_pairs_sha1' % key]), key
        in_ids = rb.outids.cpu().numpy()
