"""End to end: a whole training iteration with the fused target / loss / glue kernels equals the same
iteration through the dense tensor formulations (detmatch_amd.fused.ENABLED = False) — the formulations
that the reference-generated goldens pin piece by piece on the CPU (tests/test_pcdet_torch_golden.py,
tests/test_ssl_host.py).  Same seed -> same random keys on both paths; every logged loss within 1e-3
relative (north_star's tolerance), the accumulated gradient within 1e-3 of its norm."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(make, fused_on, seed):
    from detmatch_amd import fused
    fused.ENABLED = fused_on
    try:
        wl = make()
        torch.manual_seed(seed)
        wl.step()
        torch.cuda.synchronize()
        log = {k: float(v) for k, v in wl.last_log.items()}
        grad = wl.ddp.flat.clone()
    finally:
        fused.ENABLED = True
    return log, grad


def _compare(a, b):
    (la, ga), (lb, gb) = a, b
    assert set(la) == set(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    assert torch.isfinite(ga).all() and float(ga.abs().sum()) > 0
    assert float((ga - gb).norm() / gb.norm()) < 1e-3


def test_pvrcnn_iteration_fused_equals_tensor_formulation(dev):
    from detmatch_amd import fused, synth
    from detmatch_amd.pcdet.workload import PVRCNNTrainWorkload
    res = []
    for on in (True, False):
        fused.ENABLED = on
        try:
            wl = PVRCNNTrainWorkload([synth.lidar_frame(i) for i in range(2)], dev)
            torch.manual_seed(7)
            wl.step()
            torch.cuda.synchronize()
            res.append(({k: float(v) for k, v in wl.last_out.items()}, wl.ddp.flat.clone()))
        finally:
            fused.ENABLED = True
    assert 'loss' in res[0][0]
    _compare(res[0], res[1])


@pytest.mark.parametrize('recipe', [None, 'confthr_pvrcnn'])
def test_ssl_iteration_fused_equals_tensor_formulation(dev, recipe):
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    make = lambda: DetMatchTrainWorkload(2, dev, ssl_cfg=recipe)
    a = _run(make, True, 5)
    b = _run(make, False, 5)
    if recipe is None:
        assert any('hung' in k for k in a[0])                  # the matching chain reports its counts
    _compare(a, b)
