"""BASELINE.json configs[4], geometry leg: the Waymo-SHAPED synthetic frame (full 360 deg sweep,
~200 k points, range +-75.2 m x [-2, 4), voxel [0.1, 0.1, 0.15] -> grid 1504 x 1504 x 40, sparse shape
[41, 1504, 1504], max_voxels 150000, image 1280 x 1920) through the hot path in fp32:
voxelize + all 8 rulebooks + the 12 sparse convolutions against the oracle, then full DetMatch
iterations at bs = 1 in fp32 AND in the mixed-precision mode that config asks for (cfg.fp16 of the reference ->
detmatch_amd/precision.py: bf16 multiplicands with fp32 accumulation in the dense and sparse GEMMs), whose
logged losses must agree with the fp32 run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SHAPE = [41, 1504, 1504]


def test_waymo_shaped_voxelize_rulebooks_convs_vs_oracle(orc, dev):
    from detmatch_amd import synth, voxel
    from detmatch_amd.spconv import ops
    from test_oracle_spconv import LAYERS, layer_weight
    f = synth.lidar_frame(3, full360=True)
    pts = f['points']
    assert 150_000 < len(pts) < 320_000
    v, c, n, mean, counts = voxel.voxelize_batch([torch.from_numpy(pts).to(dev)], synth.WAYMO_VOXEL,
                                                 synth.WAYMO_RANGE, 5, 150000)
    ov, oc, on = orc.hard_voxelize(pts, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
    assert np.array_equal(c[:, 1:].cpu().numpy(), oc) and np.array_equal(n.cpu().numpy(), on)
    assert 30_000 < len(on) <= 150_000
    idx = c.cpu().numpy()
    x_o = mean.cpu().numpy()
    x_d, cur, shape, books_o, books_d = mean, c, SHAPE, {}, {}
    total_pairs = 0
    for li, (key, subm, cin, cout, ks, st, pd) in enumerate(LAYERS):
        if key not in books_o:
            books_o[key] = orc.get_indice_pairs(idx, 1, shape, ks, st, pd, subm=subm, sort_out=True)
            books_d[key] = ops.build_rulebook(cur, 1, shape, ks, st, pd, 1, subm)
            o, p, num, osh = books_o[key]
            rb = books_d[key]
            assert rb.out_shape == osh and np.array_equal(rb.indice_num.cpu().numpy(), num)
            assert np.array_equal(rb.outids.cpu().numpy(), o if not subm else idx)
        o, p, num, osh = books_o[key]
        rb = books_d[key]
        total_pairs += int(num.sum())
        w = layer_weight(li, ks, cin, cout) * 3.0
        y_o = orc.indice_conv(x_o, w.reshape(-1, cin, cout), p, num, len(o) if not subm else len(idx), subm=subm)
        y_d = ops.indice_conv(x_d, torch.from_numpy(w).to(dev), rb.indice_pairs, rb.indice_num, rb.n_out,
                              False, subm)
        scale = max(1.0, float(np.abs(y_o).max()))
        np.testing.assert_allclose(y_d.cpu().numpy(), y_o, rtol=1e-4, atol=1e-5 * scale)
        x_o = np.maximum(y_o, 0)
        x_d = torch.from_numpy(x_o).to(dev)
        if not subm:
            idx, cur, shape = o, rb.outids, osh
    assert total_pairs > 1_500_000


def test_waymo_shaped_full_iteration(dev):
    """One DetMatch iteration (2D + 3D teacher-student, bs = 1 labeled + 1 unlabeled) on the
    Waymo-shaped geometry: 188 x 188 BEV map, 1280 x 1920 image."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(1, dev, seed=11, profile='waymo')
    det = wl.model.student.detector_3d
    assert [int(v) for v in det.model.dataset.grid_size] == [1504, 1504, 40]
    l1 = wl.step()
    l2 = wl.step()
    assert torch.isfinite(l1) and torch.isfinite(l2)
    log = wl.runner.log_buffer
    for k in ('sup.sup_3d.loss', 'sup.stu.loss_rpn_cls', 'ssl.unlab.hard_pseudo_3d.loss'):
        assert k in log and all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in log[k]), k


def test_waymo_shaped_mixed_precision_iteration(dev):
    """configs[4] as the reference would run it (fp16 mixed precision): the same iteration in the mixed mode.
    Tolerances: the supervised losses are smooth functions of the network outputs for fixed targets — bf16
    multiplicands (relative 2^-8 per product, fp32 accumulation) through ~30 layers move them by < 3 %; the
    pseudo-label losses additionally depend on thresholded teacher outputs (discrete), so only their scale is
    compared."""
    from detmatch_amd import precision
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    logs = []
    for on in (False, True):
        with precision.mixed_precision(on):
            wl = DetMatchTrainWorkload(1, dev, seed=11, profile='waymo')
            assert precision.mixed() == on
            wl.step()
            torch.cuda.synchronize()
            logs.append({k: float(torch.as_tensor(v[-1]).float().mean()) for k, v in wl.runner.log_buffer.items()})
            del wl
    assert not precision.mixed()
    a, b = logs
    for k in ('sup.sup_3d.loss', 'sup.stu.loss_rpn_cls', 'sup.stu.loss_cls'):
        assert np.isfinite(b[k]) and abs(a[k] - b[k]) <= 3e-2 * abs(a[k]) + 1e-4, (k, a[k], b[k])
    assert any(abs(a[k] - b[k]) > 0 for k in a if 'loss' in k)           # the mode is really on
    for k in ('ssl.unlab.hard_pseudo_3d.loss', 'loss'):
        assert np.isfinite(b[k]) and 0.5 * abs(a[k]) - 1e-3 <= abs(b[k]) <= 2.0 * abs(a[k]) + 1e-3, (k, a[k], b[k])
