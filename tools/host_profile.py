"""cProfile of the host side of 5 DetMatch iterations: top functions by own time and by cumulative
time (where the Python / launch overhead of the step goes; blocking read-backs show up as `tolist` /
`item` / `nonzero`).

    python tools/host_profile.py
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '6')   # one hardware queue per HIP stream (detmatch_amd/__init__.py), before the runtime comes up
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
