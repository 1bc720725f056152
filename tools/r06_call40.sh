#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_sgd; mkdir -p $O
timeout 900 python -m pytest tests/test_ssl_gpu.py -q -m gpu -k "optimizer or iteration" 2>&1 | grep -E "passed|failed|^FAILED" | tail -4 | tee $O/tests.txt
python3 - <<'PY' 2>/dev/null | tee $O/sgd_us.txt
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES','4')
import torch, ctypes
from detmatch_amd import _lib
n = 41_000_000
dev = torch.device('cuda', 0)
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev); b = torch.randn(n, device=dev)
L = _lib.lib()
def run():
    _lib.check(L.dm_sgd_step_f32(_lib.ptr(p), _lib.ptr(g), _lib.ptr(b), n, 0.01, 0.9, 0.0, 1e-4, 0, None, _lib.stream()), 'sgd')
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('dm_sgd_step_f32, 41 M parameters with momentum: %.1f us = %.2f TB/s (20 B per parameter)' % (us, n * 20 / us / 1e6))
PY
export DM_BENCH_WATCHDOG=0
for round in 1 2 3; do timeout 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  %.2f ms' % d['ms_per_step'])"; done | tee $O/bench.txt
