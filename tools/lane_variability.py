"""Run-to-run variability of the accumulated gradient after k iterations: serial vs serial, and
serial vs two-lane execution (a race would show up as a much larger serial-vs-lanes difference)."""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys

import torch

sys.path.insert(0, '.')
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload  # noqa: E402


def run(lanes, steps):
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    wl.model.two_lanes = lanes
    for it in range(steps):
        torch.manual_seed(321 + it)
        wl.step()
    torch.cuda.synchronize()
    return wl.ddp.flat.clone(), float(wl.last_log['loss'])


for steps in (1, 2, 3):
    a, la = run(False, steps)
    b, lb = run(False, steps)
    c, lc = run(True, steps)
    d, ld = run(True, steps)
    rel = lambda x, y: float((x - y).norm() / y.norm())
    print('steps %d: serial-serial %.2e  lanes-lanes %.2e  serial-lanes %.2e %.2e | loss %.5f %.5f %.5f %.5f'
          % (steps, rel(a, b), rel(c, d), rel(a, c), rel(b, d), la, lb, lc, ld))
