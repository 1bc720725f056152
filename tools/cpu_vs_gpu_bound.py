"""Is the DetMatch step limited by the host (launch rate) or by the device?  Per step: wall time,
CPU time of the main thread (time.thread_time: excludes time blocked in hipStreamSynchronize /
read-backs) and CPU time of the whole process (adds the autograd backward thread).  A main-thread +
backward-thread CPU time close to the wall time means the host is the bottleneck.

    python tools/cpu_vs_gpu_bound.py [detmatch|confthr]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402


def thread_cpu():
    """{tid: (name, cpu seconds)} of every thread of this process (/proc/self/task)."""
    out = {}
    tck = os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            with open('/proc/self/task/%s/stat' % tid) as fh:
                f = fh.read()
            name = f[f.index('(') + 1:f.rindex(')')]
            rest = f[f.rindex(')') + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tck)
        except OSError:
            pass
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'detmatch'
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    dev = torch.device('cuda', 0)
    wl = DetMatchTrainWorkload(2, dev, ssl_cfg='confthr_pvrcnn' if which == 'confthr' else None)
    for _ in range(5):
        wl.step()
    torch.cuda.synchronize()
    n = 20
    th0 = thread_cpu()
    w0, t0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
    for _ in range(n):
        wl.step()
    t_launch_done = time.perf_counter()
    torch.cuda.synchronize()
    w1, t1, p1 = time.perf_counter(), time.thread_time(), time.process_time()
    print('%s: wall %.1f ms/step | main-thread CPU %.1f ms | process CPU (all threads) %.1f ms | '
          'host finished issuing %.1f ms before the device' % (
              which, (w1 - w0) / n * 1e3, (t1 - t0) / n * 1e3, (p1 - p0) / n * 1e3, (w1 - t_launch_done) * 1e3))
    th1 = thread_cpu()
    rows = sorted(((c - th0.get(tid, (nm, 0.0))[1]) / n * 1e3, nm, tid) for tid, (nm, c) in th1.items())
    print('  CPU ms per step by thread: ' + ', '.join('%s[%d] %.1f' % (nm, tid, ms) for ms, nm, tid in rows[::-1] if ms >= 0.5))

if __name__ == '__main__':
    main()
