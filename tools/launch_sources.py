"""Who launches the small kernels?  One profiled DetMatch step (torch profiler with Python stacks):
kernel launches aggregated by the innermost detmatch_amd frame (forward / host code) or by the autograd
node (backward thread), with the most frequent kernel of each source.

    python tools/launch_sources.py [top]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        wl.step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])

    def source(e):
        node = e
        while node is not None:
            if node.name.startswith('autograd::engine::evaluate_function'):
                return 'bwd ' + node.name.split(': ', 1)[-1]
            node = node.cpu_parent
        for fr in (e.stack or []):
            m = re.search(r'(detmatch_amd/[\w/]+\.py)\((\d+)\): (\w+)', fr)
            if m:
                return '%s:%s %s' % (m.group(1).replace('detmatch_amd/', ''), m.group(2), m.group(3))
        node = e
        while node is not None:
            for fr in (node.stack or []):
                m = re.search(r'(detmatch_amd/[\w/]+\.py)\((\d+)\): (\w+)', fr)
                if m:
                    return '%s:%s %s' % (m.group(1).replace('detmatch_amd/', ''), m.group(2), m.group(3))
            node = node.cpu_parent
        return '?'

    total = 0
    for e in prof.events():
        if not e.kernels:
            continue
        src = source(e)
        a = agg[src]
        for k in e.kernels:
            a[0] += 1
            a[1] += k.duration
            a[2][re.sub(r'<.*', '', k.name.replace('void ', '').replace('at::native::', ''))[:40]] += 1
            total += 1
    print('launches in the step: %d' % total)
    print('%7s %9s  %-58s %s' % ('count', 'gpu us', 'source', 'most frequent kernels'))
    for src, (n, t, ks) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print('%7d %9.0f  %-58s %s' % (n, t, src[:58], ', '.join('%s x%d' % kv for kv in ks.most_common(2))))


if __name__ == '__main__':
    main()
