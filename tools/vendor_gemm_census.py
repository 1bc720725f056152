"""Which vendor GEMMs (aten::mm / addmm / bmm / baddbmm) are left in one DetMatch iteration, with shapes, the stream lane
and the Python frame that issued them (torch profiler, CPU activities with stacks).

    python tools/vendor_gemm_census.py
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from detmatch_amd import _lib  # noqa: E402
from detmatch_amd.pcdet.workload import DetMatchTrainWorkload  # noqa: E402

OPS = {'aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::addbmm'}


def main():
    wl = DetMatchTrainWorkload(2, torch.device('cuda', 0))
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    fc0, own0, t0 = _lib.FC_GEMM_CALLS[0], _lib.OWN_LINEAR_CALLS[0], _lib.BLAS_TURNS[0]
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        wl.step()
        torch.cuda.synchronize()
    rows = collections.Counter()
    for e in prof.events():
        if e.name in OPS:
            frame = next((s for s in (e.stack or []) if 'detmatch_amd' in s), (e.stack or ['?'])[0])
            rows[(e.name, str([tuple(s) for s in (e.input_shapes or []) if s]), frame.strip()[:110])] += 1
    print('one iteration: %d vendor GEMM calls, %d turns; %d dm_fc_gemm launches, %d conv-GEMM linears'
          % (sum(rows.values()), _lib.BLAS_TURNS[0] - t0, _lib.FC_GEMM_CALLS[0] - fc0, _lib.OWN_LINEAR_CALLS[0] - own0))
    for (name, shapes, frame), n in sorted(rows.items(), key=lambda kv: -kv[1]):
        print('%3d x %-12s %-60s %s' % (n, name, shapes, frame))


if __name__ == '__main__':
    main()
