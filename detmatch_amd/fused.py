"""One switch for the fused target / loss / glue kernels.

Every chain of the DetMatch step that round 2 turned into a device kernel (anchor and proposal target
assignment, RoI / point-head / RPN / bbox-head losses, RPN proposals, BEV interpolation, the Hungarian
cost matrix) keeps its dense tensor formulation — the one the reference-generated goldens pin on the
CPU — next to the kernel.  `ENABLED = False` sends CUDA tensors through those formulations instead
(tests/test_fused_end_to_end_gpu.py compares whole training iterations both ways); the product never
clears it.  Convolutions, sparse convolutions, NMS, voxelization etc. have no tensor formulation and are
not affected.
"""
import os

# DM_FUSED=0: A/B measurements of the tensor formulations (tools/); never set by the product
ENABLED = os.environ.get('DM_FUSED', '1') != '0'


def on(flag=None):
    """`flag` = the call's explicit choice (True / False) or None for the process-wide setting."""
    return ENABLED if flag is None else bool(flag)
