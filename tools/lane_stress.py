"""Stress of the multi-stream lanes: N DetMatch iterations per mode; every logged scalar of every
iteration must be finite and the loss must stay in the range the serial run spans (a missing stream
dependency shows up as garbage sooner or later).

    python tools/lane_stress.py [steps]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402


def run(mode, steps):
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    wl.model.two_lanes, wl.model.lane_mode = False, mode
    torch.manual_seed(7)
    losses, bad = [], 0
    for i in range(steps):
        loss = float(wl.step())
        log = {k: float(v) for k, v in wl.last_log.items()}
        if not all(math.isfinite(v) for v in log.values()) or not math.isfinite(loss):
            bad += 1
            print('  step %d non-finite: %s' % (i, {k: v for k, v in log.items() if not math.isfinite(v)}))
        losses.append(loss)
    return losses, bad


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    ref, bad = run(None, steps)
    print('serial   : bad %d, loss first %.3f min %.3f max %.3f last %.3f' % (bad, ref[0], min(ref), max(ref), ref[-1]))
    for mode in ('glue', 'branches'):
        l, bad = run(mode, steps)
        print('%-9s: bad %d, loss first %.3f min %.3f max %.3f last %.3f | first-step diff vs serial %.2e' % (
            mode, bad, l[0], min(l), max(l), l[-1], abs(l[0] - ref[0])))


if __name__ == '__main__':
    main()
