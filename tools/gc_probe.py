"""How much of an iteration's host time is Python's cyclic garbage collector?  gc.callbacks around 10 iterations.
    python tools/gc_probe.py"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import detmatch_amd  # noqa: F401
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload

wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
stat = {0: [0, 0.0], 1: [0, 0.0], 2: [0, 0.0]}
t0 = [0.0]


def cb(phase, info):
    if phase == 'start':
        t0[0] = time.perf_counter()
    else:
        s = stat[info['generation']]
        s[0] += 1
        s[1] += time.perf_counter() - t0[0]


gc.callbacks.append(cb)
n = 10
t = time.perf_counter()
for _ in range(n):
    wl.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / n * 1e3
gc.callbacks.remove(cb)
print('wall %.1f ms/step; thresholds %s; objects tracked %d' % (wall, gc.get_threshold(), len(gc.get_objects())))
for g in (0, 1, 2):
    print('generation %d: %.1f collections/step, %.2f ms/step' % (g, stat[g][0] / n, stat[g][1] / n * 1e3))
gc.collect()
gc.freeze()
gc.set_threshold(50000, 20, 20)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    wl.step()
torch.cuda.synchronize()
print('after gc.freeze() + thresholds (50000, 20, 20): wall %.1f ms/step' % ((time.perf_counter() - t) / n * 1e3))
