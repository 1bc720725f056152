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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import detmatch_amd  # noqa: E402,F401  (MIOpen environment, before torch touches MIOpen)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
BATCH_PER_GPU = 2
WORKLOAD = os.environ.get("DM_BENCH_WORKLOAD", "detmatch")


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


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_spconv.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of
    this same bench command, gfx950 correction applied — tools/pmc_traffic.py); None if absent."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_spconv.json')
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    stem = kernel.rstrip('>')
    for k, v in table.items():
        if k == kernel or k.startswith(stem + ','):
            return v['traffic_bytes_per_launch']
    return None


def build_workload(dev, rank):
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import (DetMatchTrainWorkload, PVRCNNTrainWorkload,
                                              Stage3DWorkload)
    if WORKLOAD in ('detmatch', 'confthr'):
        wl = DetMatchTrainWorkload(BATCH_PER_GPU, dev, seed=5000 * rank,
                                   ssl_cfg='confthr_pvrcnn' if WORKLOAD == 'confthr' else None)
        wl.frames = [synth.lidar_frame(5000 * rank + i) for i in range(BATCH_PER_GPU)]
        return wl
    frames = [synth.lidar_frame(1000 * rank + i) for i in range(BATCH_PER_GPU)]
    cls = PVRCNNTrainWorkload if WORKLOAD == 'pvrcnn' else Stage3DWorkload
    return cls(frames, dev)


def trace_launches(wl):
    """One extra (untimed) step with the launch trace on: [(ci, co, rows, kvol, P)] per
    gather-GEMM launch, in launch order."""
    from detmatch_amd.spconv import ops as sp_ops
    sp_ops.LAUNCH_TRACE = []
    try:
        wl.step()
        torch.cuda.synchronize()
        return list(sp_ops.LAUNCH_TRACE)
    finally:
        sp_ops.LAUNCH_TRACE = None


def cpu_baseline(frames):
    """The oracle (C restatement of the reference CPU path) on ONE step of the same
    sparse-conv workload, one host core."""
    import oracle
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
    oracle.build()
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    feats, coors = [], []
    for b, f in enumerate(frames):
        v, c, n = oracle.hard_voxelize(f['points'], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
        feats.append(v.sum(1) / np.maximum(n, 1)[:, None].astype(np.float32))
        coors.append(np.concatenate([np.full((len(n), 1), b, np.int32), c], 1))
    x = np.concatenate(feats)
    idx = np.concatenate(coors)
    shape = [41, 1600, 1408]
    books, acts = {}, []
    for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
        if key not in books:
            books[key] = oracle.get_indice_pairs(idx, len(frames), shape, ks, st, pd, subm=subm)
        o, p, n, osh = books[key]
        w = (rng.standard_normal((int(np.prod(ks)), cin, cout)) * 0.05).astype(np.float32)
        y = oracle.indice_conv(x, w, p, n, len(o), subm=subm)
        acts.append((x, w, p, n, subm))
        x, idx, shape = np.maximum(y, 0), o, osh
    g = np.ones_like(x)
    for (xi, w, p, n, subm) in reversed(acts):
        g, _ = oracle.indice_conv_backward(xi, w, g, p, n, subm=subm)
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit='iters/sec', cores=1, kind='port',
                sample='1 step of the sparse-conv stage only (voxelize + 8 rulebooks + 12 sparse '
                       'convs fwd+bwd, bs=%d, no BN/optimizer), oracle/dm_oracle.c, %.1f s'
                       % (len(frames), dt))


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


def joined_world(dev, backend_is_device):
    """Number of ranks that actually joined the process group: all-reduce of ones."""
    one = torch.ones(1, dtype=torch.float32, device=dev if backend_is_device else 'cpu')
    dist.all_reduce(one)
    return int(round(float(one.item())))


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
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
    if os.environ.get('DM_CUDNN_BENCHMARK'):     # tools/miopen_tune.sh: exhaustive MIOpen find
        torch.backends.cudnn.benchmark = True
    from detmatch_amd import _lib
    wl = build_workload(dev, rank)
    if world > 1:
        wl.enable_ddp()

    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    _lib.lib().dm_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = _lib.profile_records()
    _lib.lib().dm_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # one extra untimed step with the launch trace on — on EVERY rank (a step contains the gradient
    # and log all-reduces; a rank stepping alone would dead-lock the others)
    per_step = trace_launches(wl)   # [(ci, co, rows, kvol, P)] in launch order
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events from the timed region) ----
        gg = [r for r in recs if r[0] == 0]
        roof = None
        # the synthetic batch is the same every step, so launch j of a step always sees the same
        # rulebook; data-dependent launches (pseudo-label dependent) would break the 1:1 mapping
        sig = [(r[1], r[2], r[4], r[5]) for r in gg]
        want = [(a, b, c, d) for a, b, c, d, _ in per_step] * args.steps
        if per_step and sig == want:
            groups = {}
            for j, r in enumerate(gg):
                ci, co, rows, kvol, P = per_step[j % len(per_step)]
                name = ('spconv_gg<%d,%d,%d>' % (r[1], r[2], r[3]) if r[3] else
                        'spconv_gr<%d,%d>' % (r[1], r[2]))
                g = groups.setdefault(name, dict(ms=0.0, bytes=0.0, launches=0))
                g['ms'] += r[7]
                g['bytes'] += gg_bytes(P, ci, co, kvol, rows)
                g['launches'] += 1
            name, g = max(groups.items(), key=lambda kv: kv[1]['ms'])
            achieved = g['bytes'] / (g['ms'] * 1e-3) / 1e9
            tot_ms = sum(v['ms'] for v in groups.values())
            tot_b = sum(v['bytes'] for v in groups.values())
            roof = dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=pmc_traffic(name), kernel=name,
                        avg_us=round(g['ms'] / g['launches'] * 1e3, 2), launches=g['launches'],
                        bytes_per_launch=int(g['bytes'] / g['launches']),
                        all_spconv_gg=dict(achieved=round(tot_b / (tot_ms * 1e-3) / 1e9, 1),
                                           us_per_step=round(tot_ms / args.steps * 1e3, 1)))
        # weak scaling: every rank steps through its own (2 labeled + 2 unlabeled) batch, so the
        # whole-job rate is world x steps / time (per-GPU-batch iterations per second, all ranks)
        out = dict(metric='train iters/sec', value=round(world * args.steps * 1.0 / dt, 3), unit='iters/sec',
                   n_gpus=joined, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=wl.describe(), batch_per_gpu=BATCH_PER_GPU,
                               value_is='iterations of one per-GPU batch, summed over ranks',
                               global_batch=BATCH_PER_GPU * world,
                               parallelism='dp%d' % world),
                   roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(wl.frames)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
