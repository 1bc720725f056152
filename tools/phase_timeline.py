"""Host vs device timeline of one DetMatch iteration: for every SSL module, backward and the optimizer
step, when the host entered / left it and when the device reached the same points (HIP events on
the main stream).  `lag` = device time of the exit event - host time of the exit: small (< 0.1 ms)
means the device was waiting for the host there (host-bound), large means queued work (device-bound).

    python tools/phase_timeline.py
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402

MARKS = []


def wrap(obj, attr, label):
    fn = getattr(obj, attr)

    def inner(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        t1 = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        MARKS.append((label, t0, t1, e0, e1))
        return out
    setattr(obj, attr, inner)


def main():
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    dev = torch.device('cuda', 0)
    profile = os.environ.get('DM_BENCH_PROFILE', 'kitti')
    wl = DetMatchTrainWorkload(1 if profile == 'waymo' else 2, dev, profile=profile)
    m = wl.model
    for tag, lst in (('lab', m.lab_ssl_modules), ('unlab', m.unlab_ssl_modules)):
        for i, mod in enumerate(lst):
            wrap(mod, 'forward', '%s.%02d.%s' % (tag, i, type(mod).__name__))
    wrap(m, '_update_teacher', 'ema')
    for h in wl.runner._hooks:
        if type(h).__name__ == 'OptimizerHook':
            wrap(h, 'after_train_iter', 'backward+clip+optimizer')
    for _ in range(6):
        wl.step()
    torch.cuda.synchronize()
    rows = []
    for rep in range(3):
        del MARKS[:]
        base_e = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        base_e.record()
        base_t = time.perf_counter()
        wl.step()
        end_t = time.perf_counter()
        torch.cuda.synchronize()
        done_t = time.perf_counter()
        rows = [(lab, (t0 - base_t) * 1e3, (t1 - base_t) * 1e3, base_e.elapsed_time(e0), base_e.elapsed_time(e1))
                for lab, t0, t1, e0, e1 in MARKS]
        print('step %d: host issued in %.1f ms, device done at %.1f ms' %
              (rep, (end_t - base_t) * 1e3, (done_t - base_t) * 1e3))
    print('%-44s %8s %8s %8s | %8s %8s %8s | %7s' % ('range', 'h.begin', 'h.end', 'h.dur', 'd.begin', 'd.end',
                                                     'd.dur', 'lag'))
    for lab, h0, h1, d0, d1 in rows:
        print('%-44s %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f | %7.2f' % (lab, h0, h1, h1 - h0, d0, d1, d1 - d0, d1 - h1))


if __name__ == '__main__':
    main()
