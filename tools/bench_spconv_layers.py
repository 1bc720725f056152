"""Per-layer timing of the sparse-conv kernels on the KITTI-shaped B=2 workload
(developer tool; GPU only).  Prints us / algorithmic GB/s per layer for forward,
input-gradient and weight-gradient launches.

    python tools/bench_spconv_layers.py [--reps 50] [--only subm3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from detmatch_amd import synth, voxel  # noqa: E402
from detmatch_amd.pcdet.workload import BACKBONE_LAYERS  # noqa: E402
from detmatch_amd.spconv import ops  # noqa: E402


def _n3(v):
    return list(v) if isinstance(v, (list, tuple)) else [v] * 3


def timed(fn, reps, kind=None):
    """kind None: wall time per call (torch events around the loop, includes host launch
    overhead); kind 0/1: mean in-library HIP-event time of the gather-GEMM / wgrad kernel."""
    from detmatch_amd import _lib
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if kind is not None:
        _lib.lib().dm_profile_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    if kind is not None:
        recs = [r for r in _lib.profile_records() if r[0] == kind]
        _lib.lib().dm_profile_enable(0)
        return sum(r[7] for r in recs) / max(len(recs), 1) * 1e3
    return e0.elapsed_time(e1) / reps * 1e3


def strided_chain_us(idx, batch, shape, ks, st, pd, rb, reps=20):
    """Device time of one strided rulebook's launch chain (count phase + fill phase back to back with
    N_out already known — what the GPU executes per build, without the host's read of N_out)."""
    from detmatch_amd import _lib
    L = _lib.lib()
    dev = idx.device
    n, kvol = idx.shape[0], rb.kvol
    ws = torch.empty((L.dm_rulebook_workspace_bytes(n, kvol),), dtype=torch.uint8, device=dev)
    n_out_dev = torch.empty((1,), dtype=torch.int32, device=dev)
    args = (_lib.ptr(idx), n, batch, _lib.ints(shape), _lib.ints(rb.out_shape), _lib.ints(ks), _lib.ints(st),
            _lib.ints(pd))
    outs = (torch.empty_like(rb.outids), torch.empty_like(rb.nbr_out), torch.empty_like(rb.nbr_in),
            torch.empty_like(rb.indice_pairs), torch.empty_like(rb.indice_num))

    def once():
        _lib.check(L.dm_rulebook_conv_count(*args, _lib.ptr(n_out_dev), _lib.ptr(ws), ws.numel(), _lib.stream()), 'count')
        _lib.check(L.dm_rulebook_conv_fill(*args, rb.n_out, *[_lib.ptr(o) for o in outs], _lib.ptr(ws), ws.numel(),
                                           _lib.stream()), 'fill')
    t = timed(once, reps)
    assert torch.equal(outs[1], rb.nbr_out) and int(n_out_dev.item()) == rb.n_out
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--only', default=None)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--wgrad-chunk", type=int, default=0)
    ap.add_argument("--wgrad-order", type=int, default=1)
    ap.add_argument("--mixed", action='store_true', help='fp32 rows, bf16 multiplicands (precision.set_mixed)')
    ap.add_argument("--half", default=None, choices=['float16', 'bfloat16'], help='16-bit storage')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    from detmatch_amd import _lib
    _lib.lib().dm_spconv_set_variant(args.variant)
    _lib.lib().dm_spconv_set_wgrad_chunk(args.wgrad_chunk)
    _lib.lib().dm_spconv_set_wgrad_chunk(-1 - args.wgrad_order)
    if args.mixed:
        from detmatch_amd import precision
        precision.set_mixed(True)
    sdt = getattr(torch, args.half) if args.half else torch.float32
    T = 2 if args.half else 4
    pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(args.batch)]
    _, coors, _, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    idx, shape = coors, [41, 1600, 1408]
    books = {}
    tot = dict(f=0.0, d=0.0, w=0.0, rb=0.0, bf=0.0, bb=0.0)
    print('%-13s %7s %7s %8s | %8s %7s | %8s %7s | %8s' % ('layer', 'N_in', 'N_out', 'P', 'fwd us',
                                                          'GB/s', 'dgrad us', 'GB/s', 'wgrad us'))
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key not in books:
            t_rb = timed(lambda: ops.build_rulebook(idx, args.batch, shape, ks, st, pd, 1, subm), 10)
            books[key] = ops.build_rulebook(idx, args.batch, shape, ks, st, pd, 1, subm)
            tot['rb'] += t_rb
            chain = ''
            if not subm:
                t_ch = strided_chain_us(idx.int().contiguous(), args.batch, shape, _n3(ks), _n3(st), _n3(pd), books[key])
                tot['ch'] = tot.get('ch', 0.0) + t_ch
                chain = '   launch chain alone %.1f us' % t_ch
            print('  rulebook %-12s %.1f us (build incl. host read of N_out)%s' % (key, t_rb, chain))
        rb = books[key]
        P = int(rb.indice_num.sum().item())
        kvol = rb.kvol
        lt = sdt if cin >= 16 else torch.float32
        x = torch.randn(rb.n_in, cin, device=dev).to(lt)
        w = (torch.randn(*ks, cin, cout, device=dev) * 0.05).to(lt)
        dy = torch.randn(rb.n_out, cout, device=dev).to(lt)
        rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
        if args.only is None or args.only == key:
            tf = timed(lambda: ops.indice_conv(x, w, rb.indice_pairs, rb.indice_num, rb.n_out,
                                               False, subm), args.reps, 0)
            bf = P * (cin + cout) * T + P * 8 + kvol * cin * cout * T + rb.n_out * cout * T
            td = tb = 0.0
            if cin >= 16:
                nbr = rb.nbr_out if subm else rb.nbr_in
                td = timed(lambda: ops._gather_gemm(dy, w, nbr, rb.n_in, cin, cout, 1,
                                                    1 if subm else 0), args.reps, 0)
                tb = P * (cin + cout) * T + P * 8 + kvol * cin * cout * T + rb.n_in * cin * T
            tw = timed(lambda: ops.indice_conv_backward(x, w, dy, rb.indice_pairs, rb.indice_num,
                                                        False, subm, need_input_grad=False),
                       args.reps, 1)
            print('%-13s %7d %7d %8d | %8.1f %7.0f | %8.1f %7.0f | %8.1f' % (
                '%s %d>%d' % (key, cin, cout), rb.n_in, rb.n_out, P, tf, bf / tf / 1e3, td,
                (tb / td / 1e3) if td else 0, tw))
            tot['f'] += tf
            tot['d'] += td
            tot['w'] += tw
            tot['bf'] += bf
            tot['bb'] += tb
        idx, shape = rb.outids, rb.out_shape
    print('TOTAL fwd %.1f us (%.0f GB/s)  dgrad %.1f us (%.0f GB/s)  wgrad %.1f us  rulebooks %.1f us '
          '(strided launch chains alone %.1f us)'
          % (tot['f'], tot['bf'] / max(tot['f'], 1e-9) / 1e3, tot['d'],
             tot['bb'] / max(tot['d'], 1e-9) / 1e3, tot['w'], tot['rb'], tot.get('ch', 0.0)))


if __name__ == '__main__':
    main()
