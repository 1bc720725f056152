import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys, torch
sys.path.insert(0, '.')
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
def run(lanes, steps=2):
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    wl.model.two_lanes = lanes
    for it in range(steps):
        torch.manual_seed(321 + it)
        wl.step()
    torch.cuda.synchronize()
    return wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}
a, la = run(False)
c, lc = run(True)
print('serial-lanes rel', float((a - c).norm() / a.norm()))
for k in sorted(la):
    if abs(la[k] - lc[k]) > 1e-4 * max(1, abs(la[k])):
        print('  %-45s %.6f %.6f' % (k, la[k], lc[k]))
