"""Which Python lines issue the aten operators of one DetMatch iteration?  A TorchDispatchMode counts every
dispatched operator by (innermost detmatch_amd frame, operator); views / metadata ops are skipped.  Forward
and everything autograd runs on the calling thread are covered (the device backward threads are not).

    python tools/op_sites.py [top]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import detmatch_amd  # noqa: E402,F401
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SKIP = ('view', 'reshape', 'alias', 'detach', 'expand', 'permute', 'transpose', 't.default', 'slice', 'select',
        'unsqueeze', 'squeeze', 'as_strided', 'size', 'stride', 'is_', 'sym_', 'unbind', 'split', 'chunk',
        '_unsafe_view', 'lift_fresh', 'empty', 'narrow', 'unfold', 'record_stream', '_local_scalar_dense')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        short = name.replace('aten.', '')
        if not any(short.startswith(s) for s in SKIP):
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=24)):
                if 'detmatch_amd' in fr.filename and 'tools' not in fr.filename:
                    site = '%s:%d %s' % (fr.filename.split('detmatch_amd/')[-1], fr.lineno, fr.name)
                    break
            self.count[(site, short)] += 1
        return func(*args, **(kwargs or {}))


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    mode = Sites()
    with mode:
        wl.step()
    torch.cuda.synchronize()
    by_site = collections.Counter()
    for (site, op), n in mode.count.items():
        by_site[site] += n
    print('dispatched (non-view) operators in one iteration, calling thread: %d' % sum(mode.count.values()))
    print('%6s  %-64s %s' % ('count', 'site', 'operators'))
    for site, n in by_site.most_common(top):
        ops = collections.Counter({op: c for (s, op), c in mode.count.items() if s == site})
        print('%6d  %-64s %s' % (n, site[:64], ', '.join('%s x%d' % kv for kv in ops.most_common(4))))


if __name__ == '__main__':
    main()
