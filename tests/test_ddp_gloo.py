"""world_size-2 gloo tests of the data-parallel gradient exchange: flat-arena
bucketed all-reduce == average of the per-rank gradients, unused parameters contribute zeros
(the reference's find_unused_parameters=True case), parameters/buffers start identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.bn = nn.BatchNorm1d(16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Linear(16, 4)      # never reached by the loss
        self.frozen = nn.Linear(4, 4)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.frozen(self.b(torch.relu(self.bn(self.a(x)))))


def _worker(rank, world, port, q, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d.parallel import FlatGradDDP
        torch.manual_seed(100 + rank)          # ranks start DIFFERENT: broadcast must fix it
        net = _Net()
        ddp = FlatGradDDP(net, bucket_bytes=256, mode=mode)   # tiny buckets -> several all-reduces
        assert len(ddp.buckets) > 2
        w0 = [torch.zeros_like(net.a.weight) for _ in range(world)]
        dist.all_gather(w0, net.a.weight.data)
        assert torch.equal(w0[0], w0[1])
        rm = [torch.zeros_like(net.bn.running_mean) for _ in range(world)]
        dist.all_gather(rm, net.bn.running_mean)
        assert torch.equal(rm[0], rm[1])
        torch.manual_seed(7 + rank)
        x = torch.randn(5, 8)
        for step in range(2):
            # this rank's own gradient, computed without touching .grad (no hooks fire)
            names = [n for n, p in net.named_parameters() if p.requires_grad]
            gl = torch.autograd.grad(ddp(x).square().mean(), [dict(net.named_parameters())[n] for n in names],
                                     allow_unused=True)
            local = {n: (g if g is not None else torch.zeros_like(dict(net.named_parameters())[n]))
                     for n, g in zip(names, gl)}
            ddp.zero_grad()
            loss = ddp(x).square().mean()
            loss.backward()      # buckets are all-reduced asynchronously while this runs
            # reference result: gather every rank's local gradient and average
            want = {}
            for n, g in local.items():
                gs = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(gs, g)
                want[n] = sum(gs) / world
            ddp.finish()
            for n, p in net.named_parameters():
                if p.requires_grad:
                    assert torch.allclose(p.grad, want[n], atol=1e-7), n
            assert float(net.unused.weight.grad.abs().sum()) == 0.0
            total = ddp.clip_grad_norm_(max_norm=1e-3)
            ref = torch.sqrt(sum((g ** 2).sum() for g in want.values()))
            assert torch.allclose(total, ref, rtol=1e-5)
            assert torch.linalg.vector_norm(ddp.flat) <= 1e-3 * (1 + 1e-4)
        q.put((rank, 'ok'))
    except Exception as e:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['collect', 'hooks'])
def test_flat_grad_ddp_world2(mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def test_flat_grad_single_process():
    """world 1 (no process group): same protocol, no collective."""
    from detmatch_amd.mm3d.parallel import FlatGradDDP
    torch.manual_seed(0)
    net = _Net()
    ddp = FlatGradDDP(net)
    ddp.zero_grad()
    ddp(torch.randn(3, 8)).sum().backward()
    ddp.finish()
    g = net.a.weight.grad
    assert g.data_ptr() >= ddp.flat.data_ptr() and g.abs().sum() > 0
    # two backward passes with a collect() in between accumulate
    x = torch.randn(3, 8)
    ddp.zero_grad()
    ddp(x).sum().backward()
    ddp.collect()
    assert net.a.weight.grad is None
    ddp(x).sum().backward()
    ddp.finish()
    two = net.a.weight.grad.clone()
    ddp.zero_grad()
    ddp(x).sum().backward()
    ddp.finish()
    assert torch.allclose(two, 2 * net.a.weight.grad, atol=1e-6)
    hk = FlatGradDDP(_Net(), mode='hooks')
    hk.module.a.weight.grad = None
    with pytest.raises(RuntimeError):
        hk.zero_grad()


# ---------------------------------------------------------------------------------------------
# Round 2 (ADVICE): data-dependent unused sets that DIFFER between ranks, and liveness.
class _Branchy(nn.Module):
    """`skip_mid` on one rank only: that rank produces no gradient for `mid` this step (what a
    batch without pseudo-labels / matches does to the SSL losses)."""

    def __init__(self):
        super().__init__()
        self.first = nn.Linear(8, 8)
        self.mid = nn.Linear(8, 8)
        self.last = nn.Linear(8, 2)
        self.never = nn.Linear(8, 2)         # no rank ever uses it

    def forward(self, x, skip_mid):
        h = torch.relu(self.first(x))
        if not skip_mid:
            h = h + self.mid(h)
        return self.last(h)


def _worker_uneven(rank, world, port, q, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d.parallel import FlatGradDDP
        torch.manual_seed(0)
        net = _Branchy()
        ddp = FlatGradDDP(net, bucket_bytes=64, mode=mode)      # one bucket per tensor or so
        assert len(ddp.buckets) >= 4
        torch.manual_seed(10 + rank)
        x = torch.randn(4, 8)
        for step in range(3):
            skip = (rank == 1) if step < 2 else True            # step 2: nobody uses `mid`
            params = dict(net.named_parameters())
            gl = torch.autograd.grad(net(x, skip).square().mean(), list(params.values()),
                                     allow_unused=True)
            want = {}
            for (n, p), g in zip(params.items(), gl):
                g = g if g is not None else torch.zeros_like(p)
                gs = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(gs, g)
                want[n] = sum(gs) / world
            ddp.zero_grad()
            net(x, skip).square().mean().backward()
            ddp.finish()                       # a rank-dependent launch order would hang / corrupt
            for n, p in net.named_parameters():
                assert torch.allclose(ddp._view[id(p)], want[n], atol=1e-7), (step, n)
            live = {n: bool(ddp.ever[ddp.index[id(p)]]) for n, p in net.named_parameters()}
            # `mid` got a gradient on rank 0 only: live on BOTH ranks; `never` stays dead
            assert live['mid.weight'] and live['first.weight'] and not live['never.weight']
            dead = ddp.dead_params()
            assert {id(p) for p in dead} == {id(net.never.weight), id(net.never.bias)}
            m = ddp.live_mask(0, ddp.flat.numel())
            off = ddp.offset[id(net.never.weight)]
            assert m is not None and int(m[off // 4]) == 0
            assert int(m[ddp.offset[id(net.mid.weight)] // 4]) == 1
        q.put((rank, 'ok'))
    except Exception:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['collect', 'hooks'])
def test_rank_dependent_unused_sets(mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def test_never_used_parameters_are_not_stepped():
    """confthr_pvrcnn builds a 2D student it never trains: the reference leaves those .grad None
    (DDP find_unused_parameters + mmcv zero_grad), so SGD weight decay must not shrink them."""
    from detmatch_amd.mm3d.parallel import FlatGradDDP
    from detmatch_amd.mm3d.runner import HybridOptimizer
    torch.manual_seed(0)
    net = _Branchy()
    ddp = FlatGradDDP(net)
    opt = HybridOptimizer([torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=0.1)])
    opt._ddp = ddp
    w_never, w_mid = net.never.weight.detach().clone(), net.mid.weight.detach().clone()
    x = torch.randn(4, 8)
    for step in range(3):
        ddp.zero_grad()
        net(x, skip_mid=step > 0).square().mean().backward()     # `mid` only used at step 0
        ddp.finish()
        opt.step()
    assert torch.equal(net.never.weight, w_never)                 # untouched
    assert not torch.equal(net.mid.weight, w_mid)                 # used once: decays ever after
    # .grad is hidden from the torch optimizer during its step only; the arena view (zeros) stays in place
    assert float(net.never.weight.grad.abs().sum()) == 0.0 and net.mid.weight.grad is not None
    assert net.never.weight not in opt.optimizers[0].state
    assert all(id(p) not in {id(q) for q in ddp.dead_params()} for p in net.mid.parameters())
    # everything live -> the mask is dropped (no per-step cost in the steady state)
    net2 = nn.Linear(4, 4)
    d2 = FlatGradDDP(net2)
    for _ in range(3):
        d2.zero_grad()
        net2(torch.randn(2, 4)).sum().backward()
        d2.finish()
    assert d2._all_live and d2.live_mask(0, d2.flat.numel()) is None


def _worker_hooks_unfused(rank, world, port, q):
    """hooks mode + a NON-fused torch optimizer + a parameter that never gets a gradient: the
    arena views must stay in place over several iterations, the unused parameter stays untouched (no
    weight decay, no state), and the dead set is the same on both ranks (rank 1 alone uses `late`)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d.parallel import FlatGradDDP
        from detmatch_amd.mm3d.runner import HybridOptimizer
        torch.manual_seed(3)
        net = _Net()
        net.late = nn.Linear(16, 4)                   # used from iteration 2 on, by rank 1 only
        ddp = FlatGradDDP(net, bucket_bytes=256, mode='hooks')
        opt = HybridOptimizer([torch.optim.AdamW([p for p in net.parameters() if p.requires_grad],
                                                 lr=1e-2, weight_decay=0.1)])
        opt._ddp = ddp                                # what enable_fused() records; no fused member here
        w_unused = net.unused.weight.detach().clone()
        w_late = net.late.weight.detach().clone()
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(rank))
        for it in range(4):
            ddp.zero_grad()                           # raised 'a gradient left the flat arena' at it == 1
            h = torch.relu(net.bn(net.a(x)))
            loss = net.frozen(net.b(h)).square().mean()
            if it >= 2 and rank == 1:
                loss = loss + net.late(h).square().mean()
            loss.backward()
            ddp.finish()
            opt.step()
            assert torch.equal(net.unused.weight, w_unused)             # no decay on a never-used parameter
            assert net.unused.weight not in opt.optimizers[0].state
            if it < 2:
                assert torch.equal(net.late.weight, w_late)
            assert net.a.weight.grad.data_ptr() == ddp._view[id(net.a.weight)].data_ptr()
            assert net.unused.weight.grad.data_ptr() == ddp._view[id(net.unused.weight)].data_ptr()
        assert not torch.equal(net.late.weight, w_late)                 # live on rank 1 -> stepped on BOTH ranks
        st = opt.optimizers[0].state[net.late.weight]
        assert int(st['step']) == 2                                     # its own step count
        ws = [torch.zeros_like(net.late.weight) for _ in range(world)]
        dist.all_gather(ws, net.late.weight.data)
        assert torch.equal(ws[0], ws[1])
        q.put((rank, 'ok'))
    except Exception:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_hooks_mode_with_unfused_optimizer_and_unused_parameter():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hooks_unfused, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def _worker_exchange(rank, world, port, q):
    """The gradient arena after collect / all_reduce, collect / rs_ag and hooks / rs_ag on the SAME per-rank gradients:
    bucket padding, shard offsets and the strictly ordered bucket issue of `hooks` give the same averaged gradients
    (until round 5 `rs_ag` had only ever run on a one-rank RCCL group)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d.parallel import FlatGradDDP
        results = {}
        for mode, exchange in (('collect', 'all_reduce'), ('collect', 'rs_ag'), ('hooks', 'rs_ag'), ('hooks', 'all_reduce')):
            torch.manual_seed(5)                   # same start on every rank and for every variant
            net = _Net()
            ddp = FlatGradDDP(net, bucket_bytes=200, mode=mode, exchange=exchange, broadcast=False)
            assert len(ddp.buckets) > 2
            for s_, e_ in ddp.buckets:            # every bucket cuts into `world` equal 16-byte aligned shards
                assert (e_ - s_) % (4 * world) == 0
            flats = []
            for step in range(2):
                torch.manual_seed(50 + 10 * step + rank)        # different data per rank
                x = torch.randn(6, 8)
                ddp.zero_grad()
                if step == 1:                      # two backward passes accumulate, only the last one may exchange
                    ddp.zero_grad(arm=False)
                    ddp(x[:3]).square().mean().backward()
                    ddp.collect()
                    ddp.arm()
                    ddp(x[3:]).square().mean().backward()
                else:
                    ddp(x).square().mean().backward()
                ddp.finish()
                flats.append(ddp.flat.clone())
            results[(mode, exchange)] = flats
            # liveness flags ride in the last bucket: summed over the ranks, every live parameter was seen by both
            assert float(ddp.used.max()) == float(world) and bool(ddp.ever.any())
        ref = results[('collect', 'all_reduce')]
        for key, flats in results.items():
            for a, b in zip(flats, ref):
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-8), key
        # and every rank holds the same arena
        for flats in results.values():
            got = [torch.zeros_like(flats[-1]) for _ in range(world)]
            dist.all_gather(got, flats[-1])
            assert torch.equal(got[0], got[1])
        q.put((rank, 'ok'))
    except Exception:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rs_ag_and_hooks_equal_collect_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


class _LazyToy(nn.Module):
    """A detector whose train_step returns the lazy log_vars of base_detector (the loss first, the logged values on demand)."""

    def __init__(self):
        super().__init__()
        self.student = nn.ModuleDict(dict(detector_3d=nn.Linear(2, 2)))
        self.iter, self.epoch = None, None

    def train_step(self, data, optimizer=None):
        from detmatch_amd.mm3d.base_detector import DetectorStepMixin
        x = data['lab_stu']
        loss, log_vars = DetectorStepMixin._parse_losses(self, {'loss_a': self.student['detector_3d'](x).square().mean(),
                                                                'metrics.b': x.mean().detach()})
        return dict(loss=loss, log_vars=log_vars, num_samples=2)


def _lazy_log_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d import runner as R
        torch.manual_seed(0)
        model = _LazyToy()
        opt = R.build_optimizer(model, {'constructor': 'HybridOptimizerConstructor',
                                        'student.detector_3d': dict(type='SGD', lr=0.01, step_interval=1)})
        run = R.IterBasedSSLRunner(model, optimizer=opt, max_iters=3)
        # no gradient clipping: the OptimizerHook then never reads the log buffer, nobody reads it on rank 1 at all
        run.register_training_hooks(lr_config=dict(policy='step', step=[]), optimizer_config=dict(grad_clip=None),
                                    log_config=dict(interval=2, hooks=[dict(type='TextLoggerHook')]))

        class Collective(R.Hook):                  # a hook with a collective of its own, registered AFTER the logger (EvalHook)
            def after_train_iter(self, runner):
                from detmatch_amd.mm3d.parallel import all_gather_object
                assert all_gather_object(runner.iter) == [runner.iter] * world
        run.register_hook(Collective())
        assert [type(h).__name__ for h in run._hooks][-2:] == ['Collective', 'TextLoggerHook']      # loggers stay last
        lab = [dict(stu=torch.full((4, 2), float(rank + 1)), img_metas=[0, 1])]
        run.run([lab, lab], [('train', 1)])
        assert not run._lazy_logs                  # settled on every rank, every iteration (one all-reduce each)
        logger = run._hooks[-1]
        assert len(logger.lines) == (1 if rank == 0 else 0)          # iteration 2 of 3; only rank 0 writes
        if rank == 0:
            assert logger.lines[0].startswith('Iter [2/3]') and 'metrics.b: 1.5000' in logger.lines[0], logger.lines
        if rank == 0:                              # a reader that exists on one rank only (a logger, bench.py)
            vals = [float(v) for v in run.log_buffer['metrics.b']]
            assert vals == pytest.approx([1.5, 1.5, 1.5])          # mean of the ranks' 1.0 and 2.0
        q.put((rank, 'ok'))
    except Exception:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_lazy_log_vars_are_settled_on_every_rank_world2():
    """runner._settle_lazy_logs: the packed all-reduce of the logged values is started by every rank at the same point of
    the iteration — never by a reader that only rank 0 has (would hang the job)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lazy_log_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def _multi_test_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from detmatch_amd.mm3d.datasets import multi_gpu_test

        class Loader(object):                     # KittiTestLoader's sharding: rank r holds frames r, r + world, ...
            def __init__(self, n):
                self.indices = list(range(rank, n, world))

            def __iter__(self):
                for i in self.indices:
                    yield dict(frame=[i])

        class Model(nn.Module):
            def forward(self, return_loss=True, rescale=False, frame=None):
                assert not return_loss and rescale and not self.training
                return [dict(frame=f, rank=rank) for f in frame]

        out = multi_gpu_test(Model(), Loader(7))
        if rank == 0:
            assert [r['frame'] for r in out] == list(range(7))             # dataset order again
            assert [r['rank'] for r in out] == [0, 1, 0, 1, 0, 1, 0]
        else:
            assert out is None
        q.put((rank, 'ok'))
    except Exception:      # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_validation_results_are_gathered_in_dataset_order_world2():
    """datasets.multi_gpu_test (mmdet multi_gpu_test for the rank-sharded validation loader): rank 0 evaluates the frames
    in dataset order, the other ranks get None."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_multi_test_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)
