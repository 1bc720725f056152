import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn, torch.nn.functional as F
from detmatch_amd.bn_relu import bn_relu_rows
dev=torch.device('cuda',0)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
for n,c in [(884736,64),(30336,32),(65536,32),(131072,16)]:
    x=torch.randn(n,c,device=dev,requires_grad=True); bn=nn.BatchNorm1d(c).to(dev); g=torch.randn(n,c,device=dev)
    def fused():
        y=bn_relu_rows(x,bn,True); y.backward(g); x.grad=None
    def ref():
        y=F.relu(bn(x)); y.backward(g); x.grad=None
    def fused_f(): bn_relu_rows(x.detach(),bn,True)
    def ref_f(): F.relu(bn(x.detach()))
    print(n,c,'fwd+bwd fused %.0f us  torch %.0f us | fwd fused %.0f torch %.0f'%(timed(fused),timed(ref),timed(fused_f),timed(ref_f)))
