"""ATen op census of one step: for the chosen ops, device time and calls grouped by input shapes."""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys, collections
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, '.')
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
want = sys.argv[1:] or ['aten::add', 'aten::add_', 'aten::clamp', 'aten::clamp_min', 'aten::mul', 'aten::copy_', 'aten::where']
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in want:
        rows[(e.key, str(e.input_shapes)[:90])][0] += e.count
        rows[(e.key, str(e.input_shapes)[:90])][1] += e.self_device_time_total
tot = collections.defaultdict(lambda: [0, 0.0])
for (k, sh), (c, t) in rows.items():
    tot[k][0] += c; tot[k][1] += t
for k, (c, t) in tot.items():
    print('%-16s calls %5d  %8.1f us' % (k, c, t))
for (k, sh), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-14s %5d %8.1f us  %s' % (k, c, t, sh))
