"""DetMatch training step, MI355X-native."""
import os as _os
import sys as _sys

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The iteration uses the
# default stream, two stream lanes (mm3d/ssl.py:_Lanes) and one side stream (_lib.aux_stream) — plus RCCL's stream under
# data parallelism: six queues give every stream its own.  Round 5 met an intermittent device dead-lock with streams
# SHARING a hardware queue (every queue waiting, no kernel running); the first same-box A/B showed none with 5 or 6
# queues, later boxes did while the geometry look-ahead still added streams — so this setting is a precaution, not the
# cure (DESIGN.md 6.R5 has the bisection; 8 queues cost 107 instead of 70 ms per iteration).  The variable is read when
# the runtime initialises, so it is set here, at import — and the lanes are only the default when that was in time.
HW_QUEUES_OK = True
if 'GPU_MAX_HW_QUEUES' in _os.environ:
    try:
        HW_QUEUES_OK = int(_os.environ['GPU_MAX_HW_QUEUES']) >= 5
    except ValueError:
        HW_QUEUES_OK = False
else:
    _t = _sys.modules.get('torch')
    if _t is not None and _t.cuda.is_initialized():
        HW_QUEUES_OK = False          # too late: the runtime is up with its default
    else:
        _os.environ['GPU_MAX_HW_QUEUES'] = '6'
