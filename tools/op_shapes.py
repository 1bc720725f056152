"""Top ATen ops of one training step by device time, grouped by input shapes (torch.profiler)."""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        wl.step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(
        sort_by='self_cuda_time_total', row_limit=70, max_name_column_width=44, max_shapes_column_width=70))


if __name__ == '__main__':
    main()
