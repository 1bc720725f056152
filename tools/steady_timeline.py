"""Host and device timeline of STEADY-STATE DetMatch iterations (no synchronisation between them, unlike
tools/phase_timeline.py): every lane call of SSL.forward_train (_Lanes.run), the geometry, every backward pass and
the optimizer hook are bracketed by HIP events on the stream they are issued on; after N back-to-back iterations the
middle one is printed — host begin/end and device begin/end of each range relative to the host's start of that
iteration, per lane.  Shows which lane the device is waiting for where, and where the host blocks.

    python tools/steady_timeline.py [n_iterations]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402

MARKS = []
ITER = [0]


class span(object):
    def __init__(self, label, lane):
        self.label, self.lane = label, lane

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        t1 = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        MARKS.append((ITER[0], self.label, self.lane, self.t0, t1, self.e0, e1))
        return False


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    from detmatch_amd.mm3d import ssl as S
    dev = torch.device('cuda', 0)
    wl = DetMatchTrainWorkload(2, dev)
    m = wl.model

    run0 = S._Lanes.run

    def run(self, module, ssl_obj, batch_dict, method='forward', lane=None):
        ln = self.lane_of(module) if lane is None else lane
        with torch.cuda.stream(self.stream(ln)):
            sp = span('%s.%s' % (type(module).__name__, method), ln)
            sp.__enter__()
        try:
            return run0(self, module, ssl_obj, batch_dict, method, lane)
        finally:
            with torch.cuda.stream(self.stream(ln)):
                sp.__exit__(None, None, None)
    S._Lanes.run = run

    def wrap(obj, attr, label, lane_fn=lambda: 0):
        fn = getattr(obj, attr)

        def inner(*a, **k):
            with span(label, lane_fn()):
                return fn(*a, **k)
        setattr(obj, attr, inner)

    def cur_lane():
        lanes = getattr(m, '_lanes', None)
        if lanes is None:
            return 0
        cur = torch.cuda.current_stream(dev)
        for i, s in enumerate(lanes.streams):
            if s == cur:
                return i
        return 9
    wrap(m, '_issue_geometry', 'geometry', cur_lane)
    wrap(m, '_share_2d_trunk', 'share_2d_trunk(issue)', lambda: 0)
    wrap(m, '_update_teacher', 'ema')
    bw0 = torch.Tensor.backward

    def backward(self, *a, **k):
        with span('backward', cur_lane()):
            return bw0(self, *a, **k)
    torch.Tensor.backward = backward
    for h in wl.runner._hooks:
        if type(h).__name__ == 'OptimizerHook':
            wrap(h, 'after_train_iter', 'optimizer hook (backward+clip+step)')
    for _ in range(6):
        wl.step()
    torch.cuda.synchronize()
    del MARKS[:]
    base_e = torch.cuda.Event(enable_timing=True)
    base_e.record()
    torch.cuda.synchronize()
    base_t = time.perf_counter()
    starts = []
    for i in range(n):
        ITER[0] = i
        starts.append(time.perf_counter())
        wl.step()
    starts.append(time.perf_counter())
    torch.cuda.synchronize()
    end = time.perf_counter()
    print('%d iterations back to back: %.2f ms per iteration (host loop), %.2f with the final drain' %
          (n, (starts[-1] - starts[0]) * 1e3 / n, (end - starts[0]) * 1e3 / n))
    # base_e was recorded before the synchronize that base_t follows: align with a second pair
    for pick in (n // 2, n // 2 + 1):
        t_it = starts[pick]
        off = (t_it - base_t) * 1e3
        print('\niteration %d (host %.2f ms long)' % (pick, (starts[pick + 1] - t_it) * 1e3))
        print('%-48s %4s %8s %8s %7s | %8s %8s %7s' % ('range', 'lane', 'h.begin', 'h.end', 'h.dur', 'd.begin', 'd.end', 'd.dur'))
        for it, lab, lane, t0, t1, e0, e1 in MARKS:
            if it != pick:
                continue
            d0, d1 = base_e.elapsed_time(e0) - off, base_e.elapsed_time(e1) - off
            print('%-48s %4d %8.2f %8.2f %7.2f | %8.2f %8.2f %7.2f' % (lab, lane, (t0 - t_it) * 1e3, (t1 - t_it) * 1e3,
                                                                     (t1 - t0) * 1e3, d0, d1, d1 - d0))


if __name__ == '__main__':
    main()
