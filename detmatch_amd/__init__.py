"""DetMatch training step, MI355X-native."""
import os as _os

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The iteration uses four
# streams — the default stream, two stream lanes (mm3d/ssl.py:_Lanes) and one side stream (_lib.aux_stream) — plus
# RCCL's under data parallelism.  Measured in round 6 (profiles/r06_hw_queues_with_rccl.txt, one rank, three lanes):
#     no process group    4 queues 64.3 ms   6 queues 64.0 ms
#     RCCL initialised    2 queues 79.3      4 queues 64.2      5: 102.3   6: 93.6   8: 92.5   12: 94.1
# — with a communicator alive, MORE than four hardware queues cost 30 ms per iteration (round 5's "precaution" of six
# queues would have cost every rank of a multi-GPU run a third of its throughput).  The entry points (bench.py,
# __graft_entry__.py, tests/conftest.py, tools/) therefore pin the runtime's default of 4 before the runtime comes up;
# the package itself only reads the setting (importing a library must not rewrite the environment of its host
# process).  (Round 5 suspected the queue count behind an intermittent dead-lock of the lanes; the cause was two vendor
# Stream-K GEMMs in flight at once — DESIGN.md 6.R6 — and is handled in _lib.blas_turn, whatever the queue count.)
try:
    HW_QUEUES = int(_os.environ.get('GPU_MAX_HW_QUEUES', '4'))
except ValueError:
    HW_QUEUES = 4
