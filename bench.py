"""bench.py — train iters/sec of the DetMatch training step on synthetic KITTI-shaped data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on rank 0 (contract in the task description): metric/value/unit, roofline
(dominant kernel, HIP events recorded by the library on the launch stream during the timed
steps) and cpu_baseline (the oracle timed on the host cores, rank 0, N=1 only).

Workload (env DM_BENCH_WORKLOAD, named exactly in `config.workload`):
  detmatch (default) the configuration the metric is quoted on: one full DetMatch iteration
                     (2D+3D teacher-student, BASELINE.json configs[3] per-GPU shape, bs=2+2/GPU);
  pvrcnn             BASELINE.json configs[1]: PV-RCNN 3D-only supervised step, bs=2;
  confthr            configs[2]: 3D-only SSL (confthr_pvrcnn);   stage3d: voxelize+backbone only.
Inputs (raw point clouds, images, GT boxes) are resident in HBM before the timed region.  Multi-GPU: the
path shards by sample (pure data parallel): every rank draws its own frames,
the only collective is the gradient all-reduce (RCCL) — weak scaling.
"""
import argparse
import json
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
WORKLOAD = os.environ.get("DM_BENCH_WORKLOAD", "detmatch")
# 'waymo' (with DM_BENCH_WORKLOAD=detmatch): BASELINE.json configs[4] — Waymo-shaped frames (~200 k points,
# 1504x1504x41 grid, 1280x1920 image), batch 1 per GPU.  The headline bench is the default 'kitti'.
PROFILE = os.environ.get("DM_BENCH_PROFILE", "kitti")
BATCH_PER_GPU = 1 if PROFILE == 'waymo' else 2
# 'bf16': mixed precision — dense-conv forward / input-gradient GEMMs with bf16 multiplicands and fp32
# accumulation (dense_conv.set_math), the counterpart of the fp16 autocast of configs[4].  NOT the
# headline arithmetic: the default is exact fp32 everywhere.
CONV_MATH = os.environ.get("DM_CONV_MATH", "fp32")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


def gg_bytes(P, ci, co, kvol, rows_out):
    """Algorithmic bytes of one gather-GEMM launch (per-kernel form):
    P*(ci+co)*4 gathered + accumulated rows, 8 B per rulebook pair, weights once,
    final output write."""
    return P * (ci + co) * 4 + P * 8 + kvol * ci * co * 4 + rows_out * co * 4


def wgrad_bytes(P, ci, co, kvol):
    """Algorithmic bytes of one weight-gradient launch: both rows of every pair once, the pair list,
    the gradient written once (backward formula minus the input-gradient half)."""
    return P * (ci + co) * 4 + P * 8 + kvol * ci * co * 4


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak


def pmc_traffic(kernel):
    """(HBM-side bytes per launch of `kernel`, the committed file they come from): rocprofv3 PMC passes of this
    same bench command (FETCH_SIZE and WRITE_SIZE in separate --pmc runs, gfx950 correction applied —
    tools/pmc_traffic.py, tools/collect_profiles.sh).  NOT measured in this run: the counters need the profiler."""
    for name in ('r05_pmc_spconv.json', 'r04_pmc_spconv.json', 'r03_pmc_spconv.json', 'r02_pmc_spconv.json', 'r01_pmc_spconv.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        table = json.load(open(path))
        stem = kernel.rstrip('>')
        for k, v in table.items():
            if k == kernel or k.startswith(stem + ','):
                return v['traffic_bytes_per_launch'], 'profiles/' + name
    return None, None


ISOLATED_RECS = []      # dm_profile records of the one-stream, op-by-op extra step(s) (other_kernel_groups)
ISOLATED_STEPS = 3      # how many of them (30 launches of the roofline kernel each)


def other_kernel_groups(wl):
    """One extra (untimed) step with HIP events around every dense-convolution launch and every FPS launch:
    the kernels that decide the step time, beside the sparse group BASELINE's metric names.
    dense_conv: flops 2 M N K of every lattice-GEMM / weight-gradient launch over the summed event time, against
    the fp32 matrix peak (the arithmetic is fp32-class whichever instruction serves it); fps: compulsory bytes
    N * 12 per launch — the kernel is a latency chain of npoint dependent rounds, reported as such."""
    from detmatch_amd import dense_conv, pointnet2_stack as pn2
    recs = {'gemm': [], 'wgrad': [], 'fps': []}

    def timed(kind, work, fn, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        recs[kind].append((work, e0, e1))
        return out
    g0, w0, f0 = dense_conv._gemm, dense_conv._wgrad, pn2.furthest_point_sample_stack

    def gemm(x, wp, bias, y, geom, taps, residual=None):
        return timed('gemm', 2.0 * geom[0] * geom[7] * geom[8] * geom[6] * geom[15] * geom[3], g0, x, wp, bias, y, geom,
                     taps, residual)

    def wgrad(U, V, out, scale_u, geom, taps, *rest):
        return timed('wgrad', 2.0 * geom[0] * geom[1] * geom[2] * geom[3] * geom[4] * geom[9], w0, U, V, out, scale_u,
                     geom, taps, *rest)

    def fps(xyz, cnt, npoint):
        return timed('fps', float(xyz.shape[0]) * 12, f0, xyz, cnt, npoint)
    dense_conv._gemm, dense_conv._wgrad, pn2.furthest_point_sample_stack = gemm, wgrad, fps
    # kernel-quality numbers: this ONE extra step is issued op by op (no chains: a chained launch cannot be bracketed
    # from Python) and on one stream (a HIP-event pair around a launch would otherwise include the time its kernel
    # waits for CUs held by another lane) — the same kernels with the same arguments as the timed steps
    from detmatch_amd import chain
    model = getattr(wl, 'model', None)
    saved = (chain.ENABLED, getattr(model, 'two_lanes', None), getattr(model, 'lane_mode', None))
    chain.ENABLED = False
    if model is not None and saved[1] is not None:
        model.two_lanes, model.lane_mode = False, 'glue'
    from detmatch_amd import _lib
    _lib.lib().dm_profile_enable(1)        # the sparse gather-GEMMs of this step too: each kernel alone on the device
    try:
        wl.step()
        torch.cuda.synchronize()
        # (the dense-conv / FPS brackets cover the first of the extra steps; the sparse records all of them)
        dense_conv._gemm, dense_conv._wgrad, pn2.furthest_point_sample_stack = g0, w0, f0
        for _ in range(ISOLATED_STEPS - 1):
            wl.step()
        torch.cuda.synchronize()
    finally:
        ISOLATED_RECS[:] = _lib.profile_records()
        _lib.lib().dm_profile_enable(0)
        dense_conv._gemm, dense_conv._wgrad, pn2.furthest_point_sample_stack = g0, w0, f0
        chain.ENABLED = saved[0]
        if model is not None and saved[1] is not None:
            model.two_lanes, model.lane_mode = saved[1], saved[2]
    out = {}
    for kind in ('gemm', 'wgrad'):
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs[kind])
        fl = sum(w for w, _, _ in recs[kind])
        if ms > 0:
            # the ceiling of the arithmetic that actually runs: fp32 matrix instruction, bf16 matrix instruction, or
            # six bf16 products per fp32-class product (three-way split operands)
            math = dense_conv.get_math()
            peak = {'fp32_mfma': MFMA_F32_PEAK_TFLOPS, 'bf16': MFMA_BF16_PEAK_TFLOPS}.get(
                math, MFMA_BF16_PEAK_TFLOPS / 6.0)
            out['dense_conv.' + ('fwd+dgrad' if kind == 'gemm' else 'wgrad')] = dict(
                bound='mfma', launches_per_step=len(recs[kind]), ms_per_step=round(ms, 3),
                GFLOP_per_step=round(fl / 1e9, 1), achieved=round(fl / ms / 1e9, 1), peak=round(peak, 1),
                unit='TFLOP/s', frac=round(fl / ms / 1e9 / peak, 4),
                frac_of_fp32_mfma_peak=round(fl / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), math=math,
                peak_is='dense bf16 matrix peak / 6 (six bf16 products per fp32-class product)'
                if peak not in (MFMA_F32_PEAK_TFLOPS, MFMA_BF16_PEAK_TFLOPS) else 'dense matrix peak of the instruction',
                timing='HIP events around each launch (includes the launch gap of short kernels), in one extra step '
                       'issued op by op on one stream')
    if recs['fps']:
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in recs['fps'])
        by = sum(w for w, _, _ in recs['fps'])
        out['fps'] = dict(bound='latency (npoint dependent rounds on one workgroup per sample; side streams)',
                          launches_per_step=len(recs['fps']), ms_per_step=round(ms, 3),
                          achieved=round(by / ms / 1e6, 3), peak=HBM_PEAK_GBS, unit='GB/s',
                          frac=round(by / ms / 1e6 / HBM_PEAK_GBS, 6))
    return out


def build_workload(dev, rank):
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import (DetMatchTrainWorkload, PVRCNNTrainWorkload,
                                              Stage3DWorkload)
    if WORKLOAD in ('detmatch', 'confthr'):
        wl = DetMatchTrainWorkload(BATCH_PER_GPU, dev, seed=5000 * rank,
                                   ssl_cfg='confthr_pvrcnn' if WORKLOAD == 'confthr' else None,
                                   profile=PROFILE)
        wl.frames = [synth.lidar_frame(5000 * rank + i) for i in range(BATCH_PER_GPU)]
        return wl
    frames = [synth.lidar_frame(1000 * rank + i) for i in range(BATCH_PER_GPU)]
    cls = PVRCNNTrainWorkload if WORKLOAD == 'pvrcnn' else Stage3DWorkload
    return cls(frames, dev)


def trace_launches(wl):
    """One extra (untimed) step with the launch trace on: [(ci, co, rows, kvol, P)] per
    gather-GEMM launch in launch order, 'fwd' / 'dgrad' per launch, and [(ci, co, kvol, P, n_in,
    n_out)] per weight-gradient launch."""
    from detmatch_amd import chain
    from detmatch_amd.spconv import ops as sp_ops
    sp_ops.LAUNCH_TRACE, sp_ops.LAUNCH_TRACE_DIR, sp_ops.LAUNCH_TRACE_W = [], [], []
    # the trace lives in the op-by-op host path (it reads the pair count of every launch back): this one step is
    # issued without chains — same launches, same order, same arguments as the chained steps of the timed region
    saved = chain.ENABLED
    chain.ENABLED = False
    try:
        wl.step()
        torch.cuda.synchronize()
        return list(sp_ops.LAUNCH_TRACE), list(sp_ops.LAUNCH_TRACE_DIR), list(sp_ops.LAUNCH_TRACE_W)
    finally:
        chain.ENABLED = saved
        sp_ops.LAUNCH_TRACE = sp_ops.LAUNCH_TRACE_DIR = sp_ops.LAUNCH_TRACE_W = None


def _oracle_stage(oracle, frames, rng_seed=0, timings=None):
    """voxelize -> 8 rulebooks -> 12 sparse convs forward -> backward on the oracle; optional per-piece
    wall times (seconds) and §8(d) algorithmic bytes."""
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    rng = np.random.default_rng(rng_seed)
    T = timings if timings is not None else {}
    for k in ('voxelize', 'rulebook', 'conv_fwd', 'conv_bwd', 'bytes_fwd', 'bytes_bwd'):
        T.setdefault(k, 0.0)
    t0 = time.perf_counter()
    feats, coors = [], []
    for b, f in enumerate(frames):
        v, c, n = oracle.hard_voxelize(f['points'], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        feats.append(v.sum(1) / np.maximum(n, 1)[:, None].astype(np.float32))
        coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
    T['voxelize'] += time.perf_counter() - t0
    x = np.concatenate(feats)
    idx = np.concatenate(coors)
    shape = [41, 1600, 1408]
    books, acts = {}, []
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key not in books:
            t0 = time.perf_counter()
            books[key] = oracle.get_indice_pairs(idx, len(frames), shape, ks, st, pd, subm=subm)
            T['rulebook'] += time.perf_counter() - t0
        o, p, n, osh = books[key]
        kvol = int(np.prod(ks))
        w = (rng.standard_normal((kvol, cin, cout)) * 0.05).astype(np.float32)
        t0 = time.perf_counter()
        y = oracle.indice_conv(x, w, p, n, len(o), subm=subm)
        T['conv_fwd'] += time.perf_counter() - t0
        P = int(n.sum())
        T['bytes_fwd'] += gg_bytes(P, cin, cout, kvol, len(o))
        T['bytes_bwd'] += P * (2 * cin + 2 * cout) * 4 + P * 8 + 2 * kvol * cin * cout * 4 + len(x) * cin * 4
        acts.append((x, w, p, n, subm))
        x, idx, shape = np.maximum(y, 0), o, osh
    g = np.ones_like(x)
    t0 = time.perf_counter()
    for (xi, w, p, n, subm) in reversed(acts):
        g, _ = oracle.indice_conv_backward(xi, w, g, p, n, subm=subm)
    T['conv_bwd'] += time.perf_counter() - t0
    return T


def _reference_stage(frames, threads):
    """The same stage on the REFERENCE's own CPU kernels compiled in the build container (oracle/_ref: the unmodified
    voxelizer; geometry.h + reordering.cc with torch::mm_out between gather and scatter under oracle/ref_spconv_driver.cc)
    — present on the GPU box as prebuilt files only (nothing under /root/reference is read at run time).
    -> per-piece seconds, or None when the prebuilt modules are not there / do not load."""
    try:
        from oracle import build_ref
        ref = build_ref.load_ref(build_ref.SPCONV_NAME)
        if ref is None or build_ref.load_ref(build_ref.NAME) is None:
            return None
    except Exception as e:      # optional: the port is the fallback baseline
        sys.stderr.write('bench.py: oracle/_ref not usable (%s); cpu_baseline.kind = port\n' % (e,))
        return None
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    from detmatch_amd.spconv.ops import get_conv_output_size
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        T = dict(voxelize=0.0, rulebook=0.0, conv_fwd=0.0, conv_bwd=0.0)
        rng = np.random.default_rng(0)
        t0 = time.perf_counter()
        feats, coors = [], []
        for b, f in enumerate(frames):
            v, c, n = build_ref.ref_hard_voxelize(f['points'], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
            feats.append((v.sum(1) / np.maximum(n, 1)[:, None]).astype(np.float32))
            coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
        T['voxelize'] = time.perf_counter() - t0
        x = torch.from_numpy(np.concatenate(feats))
        cur = torch.from_numpy(np.concatenate(coors).astype(np.int32))
        shape, books, acts = [41, 1600, 1408], {}, []
        for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
            osh = shape if subm else get_conv_output_size(shape, ks, st, pd, [1, 1, 1])
            if key not in books:
                t0 = time.perf_counter()
                books[key] = ref.get_indice_pairs(cur, len(frames), osh, list(ks), list(st), list(pd), [1, 1, 1], bool(subm))
                T['rulebook'] += time.perf_counter() - t0
            o, p, n = books[key]
            w = torch.from_numpy((rng.standard_normal((int(np.prod(ks)), cin, cout)) * 0.05).astype(np.float32)).view(
                *ks, cin, cout)
            t0 = time.perf_counter()
            y = ref.indice_conv(x, w, p, n, o.shape[0], False, bool(subm))
            T['conv_fwd'] += time.perf_counter() - t0
            acts.append((x, w, p, n, bool(subm)))
            x, cur, shape = torch.relu(y), o, osh
        g = torch.ones_like(x)
        t0 = time.perf_counter()
        for xi, w, p, n, subm in reversed(acts):
            g, _ = ref.indice_conv_backward(xi, w, g, p, n, False, subm)
        T['conv_bwd'] = time.perf_counter() - t0
        T['total'] = T['voxelize'] + T['rulebook'] + T['conv_fwd'] + T['conv_bwd']
        return T
    except Exception as e:
        sys.stderr.write('bench.py: reference stage failed (%s); cpu_baseline.kind = port\n' % (e,))
        return None
    finally:
        torch.set_num_threads(old)


def cpu_baseline(frames, gpu_pieces=None):
    """The oracle (C restatement of the reference CPU path, oracle/dm_oracle.c) on the host cores of
    this box, per piece as BASELINE.md §4 asks: voxelize ms/frame, rulebooks ms, sparse conv forward /
    backward ms with the §8(d) algorithmic GB/s, rotated BEV IoU us/pair — at 1 thread and with one
    independent copy of the stage per core (the port is scalar C; ctypes releases the GIL).  `value` =
    sparse-conv-stage iterations per second at 1 thread; the GPU's time for the SAME pieces is printed
    beside it (`gpu_same_pieces`)."""
    import concurrent.futures
    import oracle
    oracle.build()
    t_all = time.perf_counter()
    _oracle_stage(oracle, frames[:1])                      # warm-up (page-in, first-touch)
    T = {}
    t0 = time.perf_counter()
    _oracle_stage(oracle, frames, timings=T)
    dt1 = time.perf_counter() - t0
    rng = np.random.default_rng(1)
    nb = 256
    boxes = np.concatenate([rng.uniform(0, 40, (nb, 2)), rng.uniform(-1, 1, (nb, 1)), rng.uniform(1, 4, (nb, 3)),
                            rng.uniform(-3, 3, (nb, 1))], 1).astype(np.float32)
    t0 = time.perf_counter()
    oracle.boxes_iou_bev(boxes, boxes)
    iou_us = (time.perf_counter() - t0) / (nb * nb) * 1e6
    cores = min(os.cpu_count() or 1, 16)     # bounded: the port is memory-bound well before that
    with concurrent.futures.ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda i: _oracle_stage(oracle, frames, rng_seed=i), range(cores)))
        dtn = time.perf_counter() - t0
    pieces = dict(voxelize_ms_per_frame=round(T['voxelize'] / len(frames) * 1e3, 2),
                  rulebooks_ms=round(T['rulebook'] * 1e3, 1),
                  conv_fwd_ms=round(T['conv_fwd'] * 1e3, 1), conv_bwd_ms=round(T['conv_bwd'] * 1e3, 1),
                  conv_fwd_GBps=round(T['bytes_fwd'] / T['conv_fwd'] / 1e9, 2),
                  conv_bwd_GBps=round(T['bytes_bwd'] / T['conv_bwd'] / 1e9, 2),
                  bev_iou_us_per_pair=round(iou_us, 3))
    out = dict(value=round(1.0 / dt1, 4), unit='iters/sec of the sparse-conv stage (voxelize + 8 rulebooks '
               '+ 12 sparse convs fwd+bwd, bs=%d, no BN/optimizer)' % len(frames), cores=1, kind='port',
               sample='1 warm-up + 1 timed stage at 1 thread, then %d concurrent copies (one per thread); '
                      'BEV IoU on %dx%d boxes; %.1f s in all' % (cores, nb, nb, time.perf_counter() - t_all),
               note='the port is plain C (gcc -O3 -mavx2, loops vectorised without reassociation); the REFERENCE\'s '
                    'own CPU path compiled in the build container (MKL-backed torch::mm_out between its gather and '
                    'scatter) is 2.1x faster on the convolutions forward, 3.5x backward, equal on the rulebooks and '
                    '2x slower on the voxelizer at one thread: profiles/r03_cpu_reference_vs_port.txt',
               pieces_1_thread=pieces,
               all_cores=dict(cores=cores, value=round(cores / dtn, 4),
                              note='%d independent copies of the stage in %.1f s' % (cores, dtn)))
    # the reference's own CPU kernels, when their prebuilt modules travelled with the snapshot (oracle/_ref)
    threads = min(os.cpu_count() or 1, 16)
    r1 = _reference_stage(frames, 1)
    if r1 is not None:
        r1 = _reference_stage(frames, 1)           # second pass: the first touches the voxelizer's 360 MB grid
        rn = _reference_stage(frames, threads)
        ms = lambda T: dict(voxelize_ms_per_frame=round(T['voxelize'] / len(frames) * 1e3, 2), rulebooks_ms=round(T['rulebook'] * 1e3, 1),
                            conv_fwd_ms=round(T['conv_fwd'] * 1e3, 1), conv_bwd_ms=round(T['conv_bwd'] * 1e3, 1))
        out['port'] = dict(value=out['value'], cores=1, pieces_1_thread=out.pop('pieces_1_thread'), all_cores=out.pop('all_cores'),
                           note=out.pop('note'))
        out.update(value=round(1.0 / r1['total'], 4), cores=1, kind='reference',
                   sample=out['sample'] + '; reference stage: 2 passes at 1 thread (second timed), 1 at %d threads' % threads,
                   note='oracle/_ref: the reference\'s voxelization_cpu.cpp (unmodified) and geometry.h + reordering.cc with '
                        'torch::mm_out (MKL) between gather and scatter, compiled in the build container; `port` = the C '
                        'restatement (oracle/dm_oracle.c) on the same frames',
                   pieces_1_thread=ms(r1))
        if rn is not None:
            out['all_threads'] = dict(threads=threads, value=round(1.0 / rn['total'], 4), pieces=ms(rn),
                                      note='torch.set_num_threads(%d): the per-offset GEMMs are too small for MKL to scale' % threads)
    if gpu_pieces is not None:
        out['gpu_same_pieces'] = gpu_pieces
    return out


def gpu_stage_pieces(frames, dev):
    """Device time (HIP events on the launch stream) of the same pieces as cpu_baseline."""
    from detmatch_amd import iou3d_nms, synth, voxel
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    from detmatch_amd.spconv import ops
    pts = [torch.from_numpy(f['points']).to(dev) for f in frames]

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    t_vox, vox = timed(lambda: voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000))
    _, coors, _, mean, _ = vox

    def rulebooks():
        books, cur, shape = {}, coors, [41, 1600, 1408]
        for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
            if key not in books:
                books[key] = ops.build_rulebook(cur, len(pts), shape, ks, st, pd, 1, subm)
                cur, shape = books[key].outids, books[key].out_shape
        return books
    t_rb, books = timed(rulebooks, reps=3)
    g = torch.Generator(device='cpu').manual_seed(0)
    ws = [(torch.randn(int(np.prod(ks)), cin, cout, generator=g) * 0.05).to(dev)
          for _, _, cin, cout, ks, _, _ in BACKBONE_LAYERS]
    saved = []

    def fwd():
        x = mean
        saved.clear()
        for (key, subm, cin, cout, ks, st, pd), w in zip(BACKBONE_LAYERS, ws):
            rb = books[key]
            y = ops.indice_conv(x, w, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
            saved.append((x, w, rb, subm, cin))
            x = torch.relu(y)
        return x
    t_f, xo = timed(fwd)

    def bwd():
        gq = torch.ones_like(xo)
        for x, w, rb, subm, cin in reversed(saved):
            dx, _ = ops.indice_conv_backward(x, w, gq, rb.indice_pairs, rb.indice_num, False, subm,
                                             need_input_grad=cin >= 16)
            gq = dx
    t_b, _ = timed(bwd)
    boxes = torch.cat([torch.rand(256, 2) * 40, torch.rand(256, 1) * 2 - 1, torch.rand(256, 3) * 3 + 1,
                       torch.rand(256, 1) * 6 - 3], 1).to(dev)
    t_iou, _ = timed(lambda: iou3d_nms.boxes_iou_bev(boxes, boxes))
    return dict(voxelize_ms_per_frame=round(t_vox / len(frames), 4), rulebooks_ms=round(t_rb, 3),
                conv_fwd_ms=round(t_f, 3), conv_bwd_ms=round(t_b, 3),
                bev_iou_us_per_pair=round(t_iou * 1e3 / (256 * 256), 5),
                note='HIP events on the launch stream, same synthetic frames; rulebooks include their '
                     '4 size read-backs')


def conv_math_label():
    """What arithmetic the GEMM-shaped kernels of the timed steps ran in (read back from the library, not
    from the environment): `dtype` alone would hide that the fp32 dense convolutions are bf16 splits."""
    from detmatch_amd import dense_conv, precision
    dense = {'fp32_split': 'fp32_split(3xbf16,6 products,f32 acc) on v_mfma_f32_32x32x16_bf16',
             'fp32_mfma': 'fp32 on v_mfma_f32_32x32x2_f32',
             'bf16': 'bf16 multiplicands (rounded once), f32 acc, v_mfma_f32_32x32x16_bf16'}[dense_conv.get_math()]
    sparse = ('bf16 multiplicands, f32 acc, v_mfma_f32_16x16x32_bf16' if precision.sparse_bf16() else
              {'fp32_mfma': 'fp32 on v_mfma_f32_16x16x4_f32',
               'fp32_split': 'fp32_split(3xbf16,6 products,f32 acc) on v_mfma_f32_16x16x32_bf16'}.get(
                   precision.fp32_flavour(), precision.fp32_flavour()))
    return dict(mode=CONV_MATH, dense_conv=dense, sparse_conv=sparse,
                fc_and_rows='fp32 (v_mfma_f32_32x32x2_f32 / BLAS sgemm)')


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (reference: tools/dist_train.sh:7-9 spawns
    --nproc_per_node ranks): start N fresh ranks through torch.distributed.run as a CHILD process
    (this process has not touched the GPU and never execs), relay rank 0's JSON line and exit with
    the children's status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), '--gpus', str(args.gpus),
           '--steps', str(args.steps), '--warmup', str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append('--no-cpu-baseline')
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def beat():
    """Heartbeat of the measuring child process for watched_single_gpu_run (a no-op anywhere else)."""
    path = os.environ.get('DM_BENCH_HEARTBEAT')
    if path:
        with open(path, 'a'):
            os.utime(path, None)


def watched_single_gpu_run(args):
    """`python bench.py` at one GPU: the measurement runs in a CHILD process under a watchdog.  Round 5 met an
    intermittent device dead-lock under the three stream lanes (profiles/r05_lane_hang_ab.txt: avoided, never
    understood); should a run ever wedge, the child is killed and the bench repeats ONCE in the one-lane order of
    round 4 (`DM_TWO_LANES=0`) and says so in its line — a slower honest number instead of no number.  The watchdog
    goes by PROGRESS, not by total time: the child touches a heartbeat file after every step and phase (`beat`); it is
    killed when the file is older than `DM_BENCH_WATCHDOG_S` (150 s) — or never appeared within 900 s: the first
    `import torch` on a cold box takes minutes and must not be mistaken for a dead-lock.  This process never touches
    the GPU."""
    import signal
    import subprocess
    import tempfile
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append('--no-cpu-baseline')
    stall = float(os.environ.get('DM_BENCH_WATCHDOG_S', '0')) or 150.0
    startup = max(900.0, stall)
    first = float(os.environ.get('DM_BENCH_WATCHDOG_FIRST_S', '0'))      # (tests shorten the first attempt)
    for attempt in (0, 1):
        fd, hb = tempfile.mkstemp(prefix='dm_bench_heartbeat_')
        os.close(fd)
        os.remove(hb)                 # it exists from the child's first beat on
        env = dict(os.environ, DM_BENCH_CHILD='1', DM_BENCH_HEARTBEAT=hb)
        # (test hook: the start limit stays generous — the child's `import torch` alone takes seconds on a busy host)
        lim_start, lim_stall = (max(first * 5, 10.0), first) if (attempt == 0 and first) else (startup, stall)
        if attempt:
            env.update(DM_TWO_LANES='0', DM_BENCH_NOTE='first attempt (three stream lanes) made no progress for %d s and was '
                       'killed; this line is the one-lane order of round 4' % int(first or stall))
        # own session: the child and anything it starts die together (killpg), and a SIGTERM / SIGINT that reaches THIS
        # process (`timeout N python bench.py`, the driver's limit) is forwarded instead of orphaning a child that holds —
        # or is wedged on — the GPU
        proc = subprocess.Popen(cmd, env=env, start_new_session=True)

        def kill_child():
            if proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    proc.kill()
            proc.wait()

        def on_signal(signum, frame):
            kill_child()
            if os.path.exists(hb):
                os.remove(hb)
            sys.stderr.write('bench.py: signal %d: measuring child killed\n' % signum)
            os._exit(128 + signum)

        old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
        t0 = time.monotonic()
        why = None
        try:
            while why is None:
                try:
                    rc = proc.wait(timeout=0.5)
                    return rc
                except subprocess.TimeoutExpired:
                    pass
                try:
                    idle = time.time() - os.path.getmtime(hb)
                    if idle > lim_stall:
                        why = 'no step or phase completed for %d s' % int(lim_stall)
                except OSError:
                    if time.monotonic() - t0 > lim_start:
                        why = 'the workload did not start within %d s' % int(lim_start)
            kill_child()
        finally:
            kill_child()
            for sg, h in old.items():
                signal.signal(sg, h)
            if os.path.exists(hb):
                os.remove(hb)
        print('bench.py: %s (attempt %d): killed' % (why, attempt + 1), file=sys.stderr)
        if attempt == 0 and not device_usable():
            print('bench.py: the device does not answer after the kill; not starting a second attempt', file=sys.stderr)
            return 4
    return 3


def device_usable(timeout_s=240.0):
    """After a killed attempt: can a FRESH process still run a kernel on the device?  (A short-lived child, so that this
    process stays off the GPU.)"""
    import subprocess
    if os.environ.get('DM_BENCH_DRYRUN') or os.environ.get('DM_BENCH_FAKE_HANG'):
        return True
    code = 'import torch; x = torch.ones(1024, device="cuda:0"); assert float((x + x).sum().item()) == 2048.0'
    try:
        return subprocess.run([sys.executable, '-c', code], timeout=timeout_s, start_new_session=True).returncode == 0
    except (subprocess.TimeoutExpired, OSError):
        return False


def start_stall_guard(limit_s, rank=0, what='bench.py'):
    """Ranks started by a launcher (torchrun) have no parent of ours that could watch them (watched_single_gpu_run
    covers the 1-GPU case): a thread ends the process with a diagnostic when no step has completed for `limit_s`
    seconds, so that the intermittent device dead-lock of the stream lanes fails fast and says what to
    do instead of hanging the whole job until somebody's timeout.  -> dict: set ['t'] = time.monotonic() after every
    step, ['done'] = True at the end."""
    import threading
    state = dict(t=time.monotonic(), done=False)

    def run():
        while not state['done']:
            time.sleep(min(5.0, max(0.05, limit_s / 4.0)))
            if not state['done'] and time.monotonic() - state['t'] > limit_s:
                sys.stderr.write('%s: rank %d completed no step for %.0f s - presumably the dead-lock of the three stream '
                                 'lanes (DESIGN.md 6.R5); DM_TWO_LANES=0 runs the one-lane order, which never showed it\n'
                                 % (what, rank, limit_s))
                sys.stderr.flush()
                os._exit(17)
    threading.Thread(target=run, daemon=True, name='stall-guard').start()
    return state


def joined_world(dev, backend_is_device):
    """Number of ranks that actually joined the process group: all-reduce of ones."""
    one = torch.ones(1, dtype=torch.float32, device=dev if backend_is_device else 'cpu')
    dist.all_reduce(one)
    return int(round(float(one.item())))


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    if 'WORLD_SIZE' not in os.environ and args.gpus == 1 and not os.environ.get('DM_BENCH_CHILD') \
            and not os.environ.get('DM_BENCH_DRYRUN') and os.environ.get('DM_BENCH_WATCHDOG', '1') == '1' and not profiled:
        # (never under a profiler: its preloaded library has initialised the GPU in THIS process, and a process that
        # holds the GPU must not start another program)
        sys.exit(watched_single_gpu_run(args))
    if os.environ.get('DM_BENCH_FAKE_HANG') in ('1', '2') and os.environ.get('DM_TWO_LANES') != '0':
        # test hooks of the watchdog: the first attempt never starts ('1') / stops making progress after a heartbeat ('2')
        if os.environ['DM_BENCH_FAKE_HANG'] == '2':
            beat()
        time.sleep(10 ** 6)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks'
                         % (args.gpus, world))
    if os.environ.get('DM_BENCH_DRYRUN'):
        # launcher / process-group control flow only (CPU test, gloo): no workload, no GPU
        n = 1
        if world > 1:
            dist.init_process_group(os.environ.get('DM_DIST_BACKEND', 'gloo'))
            n = joined_world(None, False)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(dict(metric='train iters/sec', value=None, n_gpus=n, dryrun=True)))
        if n != args.gpus:
            raise SystemExit('bench.py: %d of %d ranks joined' % (n, args.gpus))
        return
    if world > 1 and os.environ.get('DM_PIN_THREADS', '1') == '1' and hasattr(os, 'sched_setaffinity'):
        # the iteration is issued by one Python thread + the autograd thread per rank: give every rank its own
        # slice of the host cores (before anything touches the GPU: the runtime's helper threads inherit it)
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // world
        if per >= 2:
            os.sched_setaffinity(0, cores[local_rank * per:(local_rank + 1) * per])
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the product path)'
    # test hooks: several ranks on ONE GPU over gloo (validates the N>1 control flow on a 1-GPU box)
    if os.environ.get('DM_FORCE_DEVICE') is not None:
        local_rank = int(os.environ['DM_FORCE_DEVICE'])
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        backend = os.environ.get('DM_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    joined = 1
    if world > 1:
        joined = joined_world(dev, os.environ.get('DM_DIST_BACKEND', 'nccl') == 'nccl')
        if joined != args.gpus:
            raise SystemExit('bench.py: %d of %d ranks joined the process group' % (joined, args.gpus))
    from detmatch_amd import _lib
    if CONV_MATH != 'fp32':
        from detmatch_amd import precision
        precision.set_mixed(True)      # dense AND sparse convolutions: bf16 multiplicands, fp32 accumulate
    wl = build_workload(dev, rank)
    if world > 1:
        wl.enable_ddp()

    guard = None
    if world > 1 and float(os.environ.get('DM_BENCH_STALL_S', '300')) > 0:
        guard = start_stall_guard(float(os.environ.get('DM_BENCH_STALL_S', '300')), rank)
    beat()
    for _ in range(args.warmup):
        wl.step()
        beat()
        if guard is not None:
            guard['t'] = time.monotonic()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    _lib.lib().dm_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
        beat()
        if guard is not None:
            guard['t'] = time.monotonic()
    torch.cuda.synchronize()
    beat()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if guard is not None:
        guard['done'] = True
    recs = _lib.profile_records()
    _lib.lib().dm_profile_enable(0)
    sync_check = None
    if world > 1:
        from detmatch_amd.mm3d.parallel import all_reduce as dm_all_reduce
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dm_all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if os.environ.get('DM_BENCH_CHECK_SYNC'):
            # data parallel by construction: after the same number of steps every rank holds the same student
            # (gradients are averaged, optimizers are deterministic) and the same EMA teacher
            flat = torch.cat([p.detach().reshape(-1).float() for p in wl.model.parameters()])
            hi, lo = flat.clone(), flat.clone()
            dm_all_reduce(hi, op=dist.ReduceOp.MAX)
            dm_all_reduce(lo, op=dist.ReduceOp.MIN)
            sync_check = dict(max_abs_diff_between_ranks=float((hi - lo).abs().max()), n_values=int(flat.numel()),
                              finite=bool(torch.isfinite(flat).all()))

    # one extra untimed step with the launch trace on — on EVERY rank (a step contains the gradient
    # and log all-reduces; a rank stepping alone would dead-lock the others)
    per_step, per_dir, per_w = trace_launches(wl)
    beat()
    others = other_kernel_groups(wl)       # on every rank too (the step contains collectives)
    beat()
    if rank == 0:
        # ---- roofline of the sparse-conv kernels (HIP events from the timed region) ----
        # the synthetic batch is the same every step, so launch j of a step always sees the same
        # rulebook; data-dependent launches (pseudo-label dependent) would break the 1:1 mapping
        gg = [r for r in recs if r[0] == 0]
        wg = [r for r in recs if r[0] == 1]
        roof = None
        sig = [(r[1], r[2], r[4], r[5]) for r in gg]
        want = [(a, b, c, d) for a, b, c, d, _ in per_step] * args.steps
        if per_step and sig == want:
            groups, by_dir = {}, {}
            for j, r in enumerate(gg):
                ci, co, rows, kvol, P = per_step[j % len(per_step)]
                name = ('spconv_gr16<%d,%d>' % (r[1], r[2]) if r[3] >= 16 else
                        'spconv_gg<%d,%d,%d>' % (r[1], r[2], r[3]) if r[3] else
                        'spconv_gr<%d,%d>' % (r[1], r[2]))
                for key, table in ((name, groups), (per_dir[j % len(per_step)], by_dir)):
                    g = table.setdefault(key, dict(ms=0.0, bytes=0.0, flops=0.0, launches=0))
                    g['ms'] += r[7]
                    g['bytes'] += gg_bytes(P, ci, co, kvol, rows)
                    g['flops'] += 2.0 * P * ci * co
                    g['launches'] += 1
            if per_w and wg:
                # a record covers one layer (dm_spconv_wgrad) or all queued layers of a backward pass
                # (dm_spconv_wgrad_batch); every traced layer of a step is inside exactly one of them, so the
                # group is the steps' records against the traced layers' algorithmic bytes
                g = by_dir.setdefault('wgrad', dict(ms=0.0, bytes=0.0, flops=0.0, launches=0))
                g['ms'] += sum(r[7] for r in wg)
                g['launches'] += len(wg)
                for ci, co, kvol, P, n_in, n_out in per_w:
                    g['bytes'] += wgrad_bytes(P, ci, co, kvol) * args.steps
                    g['flops'] += 2.0 * P * ci * co * args.steps

            def rates(g):
                sec = g['ms'] * 1e-3
                return dict(GBps=round(g['bytes'] / sec / 1e9, 1), frac_hbm=round(g['bytes'] / sec / 1e9 / HBM_PEAK_GBS, 4),
                            TFLOPs=round(g['flops'] / sec / 1e12, 2),
                            frac_of_fp32_mfma=round(g['flops'] / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                            us_per_step=round(g['ms'] / args.steps * 1e3, 1), launches_per_step=g['launches'] // args.steps)
            name, g = max(groups.items(), key=lambda kv: kv[1]['ms'])
            per_bytes = g['bytes'] / g['launches']
            per_flops = g['flops'] / g['launches']
            region_us = g['ms'] / g['launches'] * 1e3
            # The kernel's own duration: its launches in the extra step(s) issued op by op on ONE stream
            # (other_kernel_groups) — begin -> end events around a dispatch that has the device to itself, which is what
            # rocprofv3's kernel trace of this command reports for it (profiles/r06_detmatch_bench_kernel_stats.csv).
            # In the timed region the three stream lanes co-schedule it with the 2D branch's convolutions and the event
            # pair there ALSO contains the time the dispatch queues for compute units (round 5, same profiled run:
            # events 43.9 us, trace 25.6 us) — kept as `timed_region_event_pairs`, not the headline.
            iso = [r for r in ISOLATED_RECS if r[0] == 0 and not r[3] and 'spconv_gr<%d,%d>' % (r[1], r[2]) == name]
            if iso:
                avg_us, n_meas, how = sum(r[7] for r in iso) / len(iso) * 1e3, len(iso), \
                    'HIP events around each launch of the kernel in %d extra step(s) issued on ONE stream right after ' \
                    'the timed region (same kernels, same arguments, same rulebooks)' % ISOLATED_STEPS
            else:
                avg_us, n_meas, how = region_us, g['launches'], 'HIP events around each launch in the timed region'
            sec = avg_us * 1e-6
            traffic, traffic_src = pmc_traffic(name)
            # BASELINE.json's metric is "spconv HBM GB/s vs roofline": the headline is the SURVEY 8(d) byte convention
            # against the HBM peak.  What really limits the kernel is matrix-pipe issue (SQ counters,
            # profiles/r01_pmc_sq_spconv_gr.txt: 14 of 33 us matrix-pipe floor, 63 % of wave cycles in issue stalls; its
            # measured HBM traffic is a third of the algorithmic bytes because neighbouring rows hit in L2) — the FLOP
            # form against the fp32 matrix peak rides along as `mfma`.
            roof = dict(bound='hbm', achieved=round(per_bytes / sec / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(per_bytes / sec / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic,
                        traffic_source=('%s (rocprofv3 --pmc passes of this command, not measured in this run)'
                                        % traffic_src) if traffic_src else None, kernel=name,
                        avg_us=round(avg_us, 2), launches=n_meas, measured=how,
                        bytes_per_launch=int(per_bytes), flops_per_launch=int(per_flops),
                        convention='SURVEY 8(d): P*(Cin+Cout)*4 + 8P + K*Cin*Cout*4 + N_out*Cout*4',
                        mfma=dict(achieved=round(per_flops / sec / 1e12, 2), peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                                  frac=round(per_flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                  bound_detail='what limits the kernel: mfma issue (fp32 v_mfma_f32_16x16x4_f32; rows packed '
                                  'by neighbour mask, tiles ~89 % full; weights re-read from L2 per (tile, offset))'),
                        timed_region_event_pairs=dict(
                            avg_us=round(region_us, 2), launches=g['launches'],
                            frac=round(per_bytes / (region_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                            note='event pairs around the same launches INSIDE the timed region: with the stream lanes the '
                                 'interval includes queueing behind co-scheduled kernels of the other lanes, so it is an '
                                 'upper bound of the dispatch time, not the dispatch time'),
                        all_spconv={k: rates(v) for k, v in by_dir.items()}, other_kernels=others)
        # weak scaling: every rank steps through its own (2 labeled + 2 unlabeled) batch, so the
        # whole-job rate is world x steps / time (per-GPU-batch iterations per second, all ranks)
        out = dict(metric='train iters/sec', value=round(world * args.steps * 1.0 / dt, 3), unit='iters/sec',
                   n_gpus=joined, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                   scaling='weak', vs_baseline=None,
                   dtype='f32' if CONV_MATH == 'fp32' else 'mixed: bf16 multiplicands + f32 accumulate in the '
                   'dense- and sparse-conv GEMMs, f32 storage and f32 everywhere else',
                   data='synthetic',
                   config=dict(workload=wl.describe(), batch_per_gpu=BATCH_PER_GPU, conv_math=conv_math_label(),
                               value_is='iterations of one per-GPU batch, summed over ranks',
                               global_batch=BATCH_PER_GPU * world,
                               parallelism='dp%d' % world),
                   roofline=roof)
        ddp = getattr(wl, 'ddp', None)
        if ddp is not None:
            # multi-GPU readiness: how gradients are exchanged, and how many host cores a rank has
            # to issue its iteration from (one Python thread + the autograd thread + the runtime's helper thread)
            out['config']['grad_exchange'] = dict(mode=ddp.mode, collective=ddp.exchange, buckets=len(ddp.buckets),
                                                  bucket_MiB=round(max(e - s_ for s_, e in ddp.buckets) * 4 / 2 ** 20, 1))
        try:
            out['config']['host_cores_per_rank'] = len(os.sched_getaffinity(0))
        except AttributeError:
            out['config']['host_cores_per_rank'] = os.cpu_count()
        model = getattr(wl, 'model', None)
        if model is not None and hasattr(model, 'lane_mode'):
            # how SSL.forward_train orders its modules over HIP streams (pcdet/workload.py)
            out['config']['stream_order'] = 'branches' if getattr(model, 'two_lanes', False) else \
                (model.lane_mode or 'serial')
            from detmatch_amd import chain
            from detmatch_amd.pcdet import detector as _det
            side = []
            if getattr(model, 'side_wgrad', False) and getattr(model, 'two_lanes', False) and getattr(getattr(wl, 'ddp', None), 'mode', None) == 'collect':
                side.append('weight-gradient halves of the chained dense / set-abstraction backward passes')
            if _det.PFE_SIDE[0]:
                side.append('key-point encoder beside the BEV backbone (forward and backward)')
            from detmatch_amd.mm3d import ssl as _ssl
            ahead = bool(_ssl._TEACHER_AHEAD and getattr(model, 'two_lanes', False) and getattr(getattr(wl, 'runner', None), 'draw_ahead', False))
            if not ahead:
                side.append('key-point FPS of all passes (one launch)')
            out['config']['side_stream'] = side
            if ahead:
                # SSL._forward_train: these wait for the previous iteration's EMA and the batch, not for its last backward
                out['config']['ahead_of_previous_tail'] = ['voxelize + rulebooks + key-point FPS of all passes (teacher lane)',
                                                           "teacher's 2D pass (2D lane)"]
            out['config']['issue'] = ('chained: one C-ABI call per static sub-graph (dm_chain_run)' if chain.ENABLED
                                      else 'op by op') + ('' if not chain.OFF else ', families off: %s' % sorted(chain.OFF))
        # pseudo-label bookkeeping of the timed steps: proves the step exercises matching (NumPreds
        # metrics of the SSL chain, mean over the timed steps)
        lb = getattr(getattr(wl, 'runner', None), 'log_buffer', None) or {}
        pl = {k: round(float(torch.stack([torch.as_tensor(v, dtype=torch.float32).cpu() for v in vals[-args.steps:]]).mean()), 3)
              for k, vals in lb.items() if 'metrics.' in k}
        if pl:
            out['config']['pseudo_labels_per_step'] = pl
        if sync_check is not None:
            out['param_sync'] = sync_check
        if os.environ.get('DM_BENCH_NOTE'):
            out['note'] = os.environ['DM_BENCH_NOTE']
        out['last_loss'] = float(wl.runner.outputs['loss'].detach()) if getattr(wl, 'runner', None) is not None and \
            getattr(wl.runner, 'outputs', None) else None
        if world == 1 and not args.no_cpu_baseline:
            beat()
            pieces = gpu_stage_pieces(wl.frames, dev)
            beat()
            out['cpu_baseline'] = cpu_baseline(wl.frames, pieces)      # (bounded: 10-30 s of host work)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
