"""cProfile of the host side of 5 DetMatch iterations: top functions by own time and by cumulative
time (where the Python / launch overhead of the step goes; blocking read-backs show up as `tolist` /
`item` / `nonzero`).

    python tools/host_profile.py
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import detmatch_amd, torch
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
for _ in range(4): wl.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): wl.step()
torch.cuda.synchronize()
pr.disable()
for key, n in (('tottime', 45), ('cumulative', 110)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue()[:16000])
