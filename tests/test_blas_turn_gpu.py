"""The one-vendor-GEMM-at-a-time rule on the GPU (two hipBLASLt Stream-K kernels of one process in flight
at the same time dead-lock the device).  (1) The token orders GEMMs of different streams behind each other and the
results are right.  (2) Census of the REAL iteration in the order that ships (three lanes): every aten GEMM of the
step — forward, autograd thread included — is issued inside a turn."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

_GEMM_OPS = {'aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::addbmm', 'aten::_scaled_mm'}


def test_turns_serialise_gemms_across_streams(dev, monkeypatch):
    from detmatch_amd import _lib
    monkeypatch.setattr(_lib, 'FC_GEMM', False)          # the VENDOR path of blas_linear (the default is the own FC GEMM)
    torch.manual_seed(0)
    x = [torch.randn(256, 27648, device=dev), torch.randn(200, 27648, device=dev)]
    w = torch.randn(256, 27648, device=dev) * 0.01
    ref = [F.linear(a, w) for a in x]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    edges0 = _lib.BLAS_TURNS[1]
    for r in range(40):
        ys = []
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                ys.append(_lib.blas_linear(x[i], w))
        for s in streams:
            s.synchronize()
        for y, want in zip(ys, ref):
            assert torch.allclose(y, want, rtol=1e-4, atol=1e-4)
    assert _lib.BLAS_TURNS[1] - edges0 >= 79          # every GEMM but the first waited for its predecessor's event


def test_every_gemm_of_the_shipped_iteration_is_inside_a_turn(dev, monkeypatch):
    from torch.profiler import ProfilerActivity, profile
    from detmatch_amd import _lib
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    monkeypatch.setenv('DM_TWO_LANES', '1')
    wl = DetMatchTrainWorkload(2, dev)
    assert wl.model.two_lanes
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    fc0 = _lib.FC_GEMM_CALLS[0]
    _lib.BLAS_TRACE[0] = True
    try:
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            wl.step()
            torch.cuda.synchronize()
    finally:
        _lib.BLAS_TRACE[0] = False
    turns, gemms = {}, []
    for e in prof.events():
        if e.name == 'dm_blas_turn':
            turns.setdefault(e.thread, []).append((e.time_range.start, e.time_range.end))
        elif e.name in _GEMM_OPS:
            gemms.append(e)
    # the FC stacks run on the library's own GEMMs (csrc/fc_gemm.hip, conv2d.hip): what is left for the vendor library
    # is a handful of odd products at most — each of them inside a turn
    assert len(gemms) <= 12 and sum(len(v) for v in turns.values()) >= len(gemms), (len(gemms), [e.name for e in gemms])
    outside = [(e.name, e.thread, [tuple(s) for s in (e.input_shapes or [])]) for e in gemms
               if not any(a <= e.time_range.start and e.time_range.end <= b for a, b in turns.get(e.thread, []))]
    assert not outside, 'vendor GEMMs issued outside _lib.blas_turn(): %s' % outside[:8]
    assert _lib.FC_GEMM_CALLS[0] - fc0 >= 60            # forward + two gradients of the student's FC stacks, the teacher's forward


def test_off_main_lane_linear_is_the_own_gemm_and_matches(dev, monkeypatch):
    """_lib.blas_linear on a stream that is not the iteration's main lane, with the FC GEMM switched off: the library's
    convolution GEMM (no vendor kernel, so no turn edge), same values and gradients as F.linear to fp32-class accuracy."""
    from detmatch_amd import _lib
    monkeypatch.setattr(_lib, 'FC_GEMM', False)
    torch.manual_seed(1)
    side = torch.cuda.Stream()
    old = _lib.MAIN_STREAM[0]
    _lib.MAIN_STREAM[0] = torch.cuda.current_stream().cuda_stream
    try:
        for m, k, n, bias in ((200, 27648, 256, False), (256, 256, 8, True), (1024, 1024, 16, True), (300, 640, 128, False)):
            x = torch.randn(m, k, device=dev, requires_grad=True)
            w = torch.nn.Parameter(torch.randn(n, k, device=dev) / k ** 0.5)
            b = torch.nn.Parameter(torch.randn(n, device=dev)) if bias else None
            g = torch.randn(m, n, device=dev)
            want = F.linear(x, w, b)
            gw = torch.autograd.grad(want, [x, w] + ([b] if bias else []), g)
            side.wait_stream(torch.cuda.current_stream())
            calls, turns = _lib.OWN_LINEAR_CALLS[0], _lib.BLAS_TURNS[0]
            with torch.cuda.stream(side):
                got = _lib.blas_linear(x, w, b)
                gg = torch.autograd.grad(got, [x, w] + ([b] if bias else []), g)
                got2 = _lib.blas_linear(x, w, b)              # second call: packed weight from the cache
            side.synchronize()
            assert _lib.OWN_LINEAR_CALLS[0] == calls + 2, (_lib.OWN_LINEAR_CALLS[0], calls)
            assert _lib.BLAS_TURNS[0] == turns, 'a vendor GEMM was issued off the main lane'
            assert got.shape == want.shape and torch.equal(got, got2), (got.shape, want.shape)
            scale = float(want.detach().abs().max())
            err = float((got.detach() - want.detach()).abs().max())
            assert err <= 1e-4 * scale, (m, k, n, err, scale)
            for which, a, c in zip('xwb', gg, gw):
                e, sc = float((a - c).abs().max()), float(c.abs().max())
                assert e <= 2e-4 * sc, (m, k, n, which, e, sc)
        # inference with an output width that is not a multiple of 4 (the teacher's 7-wide box head): own GEMM;
        # the same layer with gradients: vendor GEMM inside a turn
        x = torch.randn(256, 256, device=dev)
        w = torch.nn.Parameter(torch.randn(7, 256, device=dev) / 16)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            calls, turns = _lib.OWN_LINEAR_CALLS[0], _lib.BLAS_TURNS[0]
            with torch.no_grad():
                y = _lib.blas_linear(x, w)
            assert _lib.OWN_LINEAR_CALLS[0] == calls + 1 and _lib.BLAS_TURNS[0] == turns
            y2 = _lib.blas_linear(x, w)
            assert _lib.BLAS_TURNS[0] == turns + 1 and y2.requires_grad
        side.synchronize()
        assert float((y - F.linear(x, w).detach()).abs().max()) <= 1e-4 * float(y.abs().max())
    finally:
        _lib.MAIN_STREAM[0] = old
