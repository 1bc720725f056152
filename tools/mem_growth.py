"""Device memory of the bench workload over many iterations (allocated / reserved by the caching allocator, and the
process's host RSS): a leak in a cache keyed by stream / thread / batch identity would show as growth.
    python tools/mem_growth.py [iterations]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import os
import resource
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa
import torch
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
for h in wl.runner._hooks:
    if getattr(h, 'base_lr', None):
        h.base_lr = [lr * 1e-4 for lr in h.base_lr]
for i in range(n + 1):
    wl.step()
    if i in (20, 100) or (i % 200 == 0 and i > 0):
        torch.cuda.synchronize()
        print('iteration %5d: allocated %.1f MiB, reserved %.1f MiB, host max RSS %.1f MiB' % (
            i, torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20,
            resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
