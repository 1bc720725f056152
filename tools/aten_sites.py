"""Which Python lines pay for the aten kernels of one DetMatch iteration?  torch profiler with stacks: self device
time and launches of every aten operator, grouped by (operator, innermost detmatch_amd frame of its stack); backward
operators carry the autograd node instead of a frame.

    python tools/aten_sites.py [top]
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
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        wl.step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        dt = getattr(e, 'self_device_time_total', 0) or 0
        if dt <= 0 or not e.name.startswith('aten::'):
            continue
        site = None
        node = e
        while node is not None and site is None:
            if node.name.startswith('autograd::engine::evaluate_function'):
                site = 'bwd ' + node.name.split(': ', 1)[-1]
            for fr in (node.stack or []):
                m = re.search(r'(detmatch_amd/[\w/]+\.py)\((\d+)\): (\w+)', fr)
                if m:
                    site = '%s:%s %s' % (m.group(1).replace('detmatch_amd/', ''), m.group(2), m.group(3))
                    break
            node = node.cpu_parent
        k = (site or '?', e.name)
        agg[k][0] += 1
        agg[k][1] += dt
    total = sum(v[1] for v in agg.values())
    print('aten operators with device time: %d calls, %.2f ms' % (sum(v[0] for v in agg.values()), total / 1e3))
    by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for (site, op), (c, t) in agg.items():
        by_site[site][0] += c
        by_site[site][1] += t
        by_site[site][2][op.replace('aten::', '')] += t
    print('%7s %9s  %-58s %s' % ('calls', 'us', 'site', 'operators (us)'))
    for site, (c, t, ops) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%7d %9.1f  %-58s %s' % (c, t, site[:58], ', '.join('%s %.0f' % kv for kv in ops.most_common(4))))


if __name__ == '__main__':
    main()
