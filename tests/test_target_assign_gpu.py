"""dm_anchor_assign (csrc/anchor_assign.hip) vs the per-(class, sample) dense form of
axis_aligned_target_assigner.py:132-209 (`_assign_targets_loop`, itself pinned by the reference golden
`ah_labels` in tests/test_pcdet_torch_golden.py): labels and weights bit-exact, regression targets to
1e-6 (device logf / sqrtf vs torch's)."""
import numpy as np
import pytest
import torch

from test_target_assign import _head

pytestmark = pytest.mark.gpu


def _gt_batch(extra_random=0, seed=0):
    from detmatch_amd import synth
    rng = np.random.default_rng(seed)
    gts = []
    for s in range(2):
        f = synth.lidar_frame(s)
        lab = synth._SIM_TO_CFG_LABEL[f['gt_labels']] + 1
        g = np.concatenate([f['gt_boxes'].copy(), lab[:, None].astype(np.float32)], 1)
        if extra_random:       # pseudo-label-like crowd: many boxes, some nearly identical
            sizes = np.array([[0.8, 0.6, 1.73], [1.76, 0.6, 1.73], [3.9, 1.6, 1.56]], np.float32)
            cls = rng.integers(1, 4, extra_random)
            ex = np.concatenate([rng.uniform(0, 70, (extra_random, 1)), rng.uniform(-39, 39, (extra_random, 1)),
                                 rng.uniform(-1.2, -0.6, (extra_random, 1)), sizes[cls - 1] * rng.uniform(0.9, 1.1, (extra_random, 3)),
                                 rng.uniform(-3.1, 3.1, (extra_random, 1)), cls[:, None]], 1).astype(np.float32)
            ex[1] = ex[0]
            g = np.concatenate([g, ex])
        gts.append(g)
    gts[1] = gts[1][:max(7, extra_random)]
    gts[1][3, 7] = 0                                   # padding-like class id inside the kept range
    M = max(len(g) for g in gts)
    gt = np.zeros((3, M, 8), np.float32)               # third sample: no ground truth at all
    for k, g in enumerate(gts):
        gt[k, :len(g)] = g
    return torch.from_numpy(gt)


@pytest.mark.parametrize('extra', [0, 90])
def test_device_assignment_equals_reference_form(dev, extra):
    head = _head().to(dev)
    anchors = [getattr(head, 'anchors_%d' % i) for i in range(head._n_anchor_sets)]
    gt = _gt_batch(extra, seed=extra).to(dev)
    ta = head.target_assigner
    got = ta.assign_targets(anchors, gt)                    # device kernel
    want = ta._assign_targets_loop(anchors, gt)             # reference form, torch ops
    assert torch.equal(got['box_cls_labels'], want['box_cls_labels'].int())
    assert torch.equal(got['reg_weights'], want['reg_weights'])
    np.testing.assert_allclose(got['box_reg_targets'].cpu().numpy(), want['box_reg_targets'].cpu().numpy(),
                               rtol=1e-5, atol=1e-6)
    assert int((got['box_cls_labels'] > 0).sum()) > 50
    assert int((got['box_cls_labels'][2] != 0).sum()) == 0
    assert int((got['box_cls_labels'] == -1).sum()) > 0     # ignore band exists
