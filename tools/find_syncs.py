"""List the host<->device synchronisation points of one training step (torch sync-debug mode).

    python tools/find_syncs.py [detmatch|pvrcnn|confthr]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import collections
import sys
import traceback
import warnings

import torch

sys.path.insert(0, '.')


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'detmatch'
    dev = torch.device('cuda', 0)
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload, PVRCNNTrainWorkload
    if which == 'pvrcnn':
        wl = PVRCNNTrainWorkload([synth.lidar_frame(i) for i in range(2)], dev)
    else:
        wl = DetMatchTrainWorkload(2, dev, ssl_cfg='confthr_pvrcnn' if which == 'confthr' else None)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    sites = collections.Counter()

    def showwarning(message, category, filename, lineno, file=None, line=None):
        if 'synchroniz' not in str(message):
            return
        stack = [f for f in traceback.extract_stack() if '/detmatch_amd/' in f.filename]
        key = ' <- '.join('%s:%d' % (f.filename.split('detmatch_amd/')[-1], f.lineno) for f in stack[::-1][:4])
        sites[key] += 1
    warnings.showwarning = showwarning
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode(1)
    wl.step()
    torch.cuda.set_sync_debug_mode(0)
    print('%d synchronising calls in one step' % sum(sites.values()))
    for k, v in sites.most_common():
        print('%4d  %s' % (v, k))


if __name__ == '__main__':
    main()
