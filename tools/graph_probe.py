import sys, time, torch
sys.path.insert(0, '.')
from detmatch_amd import configs
from detmatch_amd.mm2d import FasterRCNN
dev = torch.device('cuda', 0)
cfg = configs.frcnn_kitti_model(); cfg.pop('type')
torch.manual_seed(0)
m = FasterRCNN(train_cfg=configs.frcnn_train_cfg(), test_cfg=configs.frcnn_test_cfg(), **cfg).to(dev)
x = torch.randn(2, 3, 384, 1280, device=dev)

def feat(inp):
    f = m.extract_feat(inp)
    c, r = m.rpn_head(f)
    return tuple(f) + tuple(c) + tuple(r)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

m.eval()
with torch.no_grad():
    print('eval eager ms', timeit(lambda: feat(x)))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): feat(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = feat(x)
    print('eval graph ms', timeit(lambda: g.replay()))
    ref = feat(x)
    g.replay(); torch.cuda.synchronize()
    print('eval graph max err', max(float((a - b).abs().max()) for a, b in zip(out, ref)))

# training: make_graphed_callables over a module wrapper
class Feat(torch.nn.Module):
    def __init__(self, det):
        super().__init__(); self.det = det
    def forward(self, inp):
        f = self.det.extract_feat(inp)
        c, r = self.det.rpn_head(f)
        return tuple(f) + tuple(c) + tuple(r)
m.train()
fm = Feat(m)
def train_eager():
    outs = fm(x)
    sum(o.square().mean() for o in outs).backward()
print('train eager ms', timeit(train_eager, 6))
g0 = m.neck.lateral_convs[0].conv.weight.grad.clone()
for p in m.parameters(): p.grad = None
gm = torch.cuda.make_graphed_callables(fm, (x.clone().requires_grad_(False),))
def train_graph():
    outs = gm(x)
    sum(o.square().mean() for o in outs).backward()
print('train graph ms', timeit(train_graph, 6))
for p in m.parameters(): p.grad = None
train_graph(); torch.cuda.synchronize()
g1 = m.neck.lateral_convs[0].conv.weight.grad
for p in m.parameters(): p.grad = None
train_eager(); torch.cuda.synchronize()
g2 = m.neck.lateral_convs[0].conv.weight.grad
print('grad rel err graph vs eager', float((g1 - g2).norm() / g2.norm()))
