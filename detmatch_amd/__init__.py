"""DetMatch training step, MI355X-native."""
import os as _os

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The iteration uses the
# default stream, two stream lanes (mm3d/ssl.py:_Lanes) and one side stream (_lib.aux_stream) — plus RCCL's stream under
# data parallelism; streams that share a hardware queue execute one behind the other, so the ENTRY POINTS
# (bench.py, __graft_entry__.py, tests/conftest.py, tools/) ask for six queues before the runtime comes up.  The package
# itself only reads the setting (importing a library must not rewrite the environment of its host process): with fewer
# queues than streams the lanes still give the same results, only less overlap.  (Round 5 suspected the queue count
# behind an intermittent dead-lock of the lanes; the cause was two vendor Stream-K GEMMs in flight at once —
# DESIGN.md 6.R6 — and is handled in _lib.blas_turn, whatever the queue count.)
try:
    HW_QUEUES = int(_os.environ.get('GPU_MAX_HW_QUEUES', '4'))
except ValueError:
    HW_QUEUES = 4
HW_QUEUES_OK = HW_QUEUES >= 5
