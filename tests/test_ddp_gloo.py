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
