"""The issue order of a DetMatch iteration — ONE place that decides it, used by the drop-in training entry
(`datasets.train_ssl_detector`) and by the bench / test workload (`pcdet/workload.py`), so that what is measured is what a
config-driven run gets.  Scheduling only: every switch here changes WHEN work is issued and on which HIP stream, never what
is computed (`tests/test_ssl_gpu.py` holds the equality tests for each of them).

Environment (A/B): DM_TWO_LANES=1|0, DM_LANE_MODE=glue|serial (below DM_TWO_LANES=0), DM_SHARE_2D_TRUNK=1|0, DM_LOOKAHEAD.
"""
import os


def apply_issue_order(model, ddp, runner=None):
    """`model`: the SSL detector (mm3d/ssl.py); `ddp`: its FlatGradDDP; `runner`: its IterBasedSSLRunner, once it exists
    (the function may be called twice: before the arenas are built and after the runner is).  A model that is not an SSL
    detector, or a wrapper that is not a FlatGradDDP, is left alone (one stream, one backward pass, as the reference)."""
    from .parallel import FlatGradDDP
    if not hasattr(model, '_forward_train') or not isinstance(ddp, FlatGradDDP):
        return False
    # the supervised part of the loss (and the unlabeled 2D losses) are back-propagated as soon as their modules are done
    # (d sum = sum of d; the OptimizerHook arms the gradient arena instead of zeroing it)
    model.early_backward = True
    # one backbone + FPN + RPN pass for the student's labeled and unlabeled images (mm2d/faster_rcnn.py: prefetch_trunk;
    # the OptimizerHook of the runner finishes the deferred trunk backward)
    model.share_2d_trunk = os.environ.get('DM_SHARE_2D_TRUNK', '1') == '1'
    # weight-gradient halves of the chained backward passes on the side stream (the gradients are read by ddp.collect,
    # which waits for them) — with the lanes and the collect mode only; IterBasedSSLRunner.train applies it for the
    # length of an iteration
    model.side_wgrad = True
    if ddp.mode == 'collect':
        # gradients of every early backward pass are folded into the flat arena by batched multi-tensor adds and
        # released, so autograd never accumulates tensor by tensor (neutral on the step time, -190 launches)
        model.after_partial_backward = ddp.collect
    # Stream lanes (ssl.py:_Lanes; data-flow edges of the batch dict become event waits).
    #   'branches' (DEFAULT, DM_TWO_LANES=1): student 3D / both 2D detectors / teacher 3D + glue on three HIP
    #       streams, the static sub-graphs issued as chains (chain.py): the long tails of small 3D kernels run
    #       underneath the 2D convolutions — 64-66 ms per iteration against 82 in 'glue' (round 5, same box).
    #   'glue' (DM_TWO_LANES=0): every detector pass on the caller's stream, strictly ordered, only the pseudo-label
    #       glue with its host read-backs on a side stream; DM_LANE_MODE=serial: one stream.
    # Same gradients and losses in all orders (tests/test_ssl_gpu.py).  The "device dead-lock of the lanes" of
    # round 5 was a vendor Stream-K GEMM of one lane spinning for ever next to a second one of another lane
    # (DESIGN.md 6.R6); vendor GEMMs are issued one at a time since (_lib.blas_turn), and the order is the default
    # at any world size — RCCL's stream included (profiles/r06_soak_nccl_10000.txt: 10 000 iterations, one rank, nccl).
    # In the timed region a HIP-event pair around a kernel of one lane also contains the time the dispatch queues
    # behind the other lanes' kernels; bench.py therefore takes the roofline kernel's duration from extra steps
    # issued on one stream.
    model.two_lanes = os.environ.get('DM_TWO_LANES', '1') == '1'
    mode = os.environ.get('DM_LANE_MODE', 'glue')
    model.lane_mode = None if (model.two_lanes or mode in ('serial', 'none', '0', '')) else mode
    if runner is not None and model.two_lanes and 'DM_LOOKAHEAD' not in os.environ and hasattr(runner, 'draw_ahead'):
        # With the lanes the geometry is prepared at the start of its own iteration — on the teacher lane, behind the
        # previous EMA instead of the previous backward (SSL._forward_train, which needs the batch one iteration ahead:
        # draw_ahead).  The look-ahead's extra side-stream work costs more than it hides
        # (profiles/r06_ab_step_variants.txt); it was never dangerous by itself (the round-5 dead-lock: DESIGN.md 6.R6).
        runner.lookahead = False
        runner.draw_ahead = True
    return True
