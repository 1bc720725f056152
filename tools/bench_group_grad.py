"""dm_group_rows_grad at the RoI-grid-pooling size from a REAL geometry: 2 x 128 boxes x 216 grid points querying 2 x 2 048
key points with radius 1.6 (nsample 16), C = 128 in rows of 132 floats.   DM_GRG_WGS=512 python tools/bench_group_grad.py"""
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detmatch_amd import _lib

dev = torch.device('cuda', 0)
torch.manual_seed(0)
B, R, G, NS, C, N = 2, 128, 216, 16, 128, 2048
M = R * G
src = torch.rand(B, N, 3, device=dev) * torch.tensor([70.0, 80.0, 4.0], device=dev)
ctr = src[:, torch.randint(0, N, (R,), device=dev)]                                   # boxes around key points
grid = (torch.stack(torch.meshgrid(*[torch.linspace(-0.5, 0.5, 6, device=dev)] * 3, indexing='ij'), -1).reshape(G, 3)
        * torch.tensor([4.0, 2.0, 1.6], device=dev))
q = (ctr[:, :, None, :] + grid[None, None]).reshape(B, M, 3)
d = torch.cdist(q, src)                                                               # (B, M, N)
dist, nn = d.topk(NS, dim=2, largest=False)
first = nn[:, :, :1]
idx = torch.where(dist <= 1.6, nn, first.expand_as(nn)).int().reshape(B * M, NS).contiguous()   # ball query: pad with the first hit
gout = torch.randn(B * M * NS, C + 4, device=dev)
q_cnt = torch.full((B,), M, dtype=torch.int32, device=dev)
s_cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
gf = torch.empty(B * N, C, device=dev)
L = _lib.lib()
per_box = idx.view(B * R, G * NS).long()
distinct = float(torch.stack([torch.unique(r).numel() * 1.0 for r in per_box[:32]] and [torch.tensor(float(torch.unique(r).numel())) for r in per_box[:32]]).mean())


def run():
    _lib.check(L.dm_group_rows_grad(B, B * M, C, B * N, NS, C + 4, 4, _lib.ptr(gout), _lib.ptr(idx), _lib.ptr(q_cnt), _lib.ptr(s_cnt),
                                    None, _lib.ptr(gf), _lib.stream()), 'dm_group_rows_grad')


for _ in range(3):
    run()
torch.cuda.synchronize()
ref = torch.zeros(B * N, C, device=dev, dtype=torch.float64)
flat_src = (idx.long() + (torch.arange(B * M, device=dev) // M * N)[:, None]).reshape(-1)
ref.index_add_(0, flat_src, gout[:, 4:].double())
err = float((gf.double() - ref).abs().max() / ref.abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('DM_GRG_WGS=%s: %.1f us per call = %.2f TB/s of rows read; %.0f distinct key points per box; max rel err %.1e' % (
    os.environ.get('DM_GRG_WGS', '256'), us, B * M * NS * C * 4 / us / 1e6, distinct, err))
