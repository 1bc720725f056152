"""GPU tests of the teacher-student step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_ema_matches_per_tensor_formula(dev):
    """dm_ema_update_* over the flat arenas == ssl.py:146-163 applied entry by entry, including
    BN running stats and integer buffers (float math, truncated back)."""
    import torch.nn as nn
    from detmatch_amd.mm3d.ssl import SSL, _Arena
    torch.manual_seed(0)

    def net():
        return nn.Sequential(nn.Conv2d(3, 5, 3), nn.BatchNorm2d(5), nn.Linear(7, 3)).to(dev)
    t, s = net(), net()
    with torch.no_grad():
        s[1].running_mean.normal_(); s[1].running_var.uniform_(0.5, 2)
        s[1].num_batches_tracked.fill_(1001); t[1].num_batches_tracked.fill_(10)
    ssl = SSL.__new__(SSL)
    nn.Module.__init__(ssl)
    ssl.teacher, ssl.student = t, s
    ssl.ema_params = dict(ema_decay=0.999, true_avg_rampup=True, rampup_start_decay=0.99)
    ssl.rampup_start_decay, ssl.use_student_bn_stats_for_teacher, ssl._arenas = 0.99, False, None
    ssl.iter = 17
    d = ssl._get_curr_ema_decay()
    want = {k: (s.state_dict()[k] * (1 - d) + v * d) for k, v in t.state_dict().items()}
    x = torch.randn(2, 3, 9, 9, device=dev)
    ssl._update_teacher()
    for k, v in t.state_dict().items():
        if v.dtype == torch.int64:
            assert int(v) == int(want[k].to(torch.int64)), k        # 10*d + 1001*(1-d) -> trunc
        else:
            assert torch.allclose(v, want[k], rtol=1e-6, atol=1e-7), k
    # parameters still work as module parameters after re-homing into the arena
    assert torch.isfinite(t[0](x)).all()
    ssl._update_teacher()   # second call reuses the arenas


def test_detmatch_confthr_iteration(dev):
    """3D-only SSL recipe (configs/detmatch/001/confthr_pvrcnn): two full iterations."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, dev, ssl_cfg='confthr_pvrcnn')
    t0 = wl.model.teacher.detector_3d.model.backbone_3d.conv_input[0].weight.detach().clone()
    s0 = wl.model.student.detector_3d.model.backbone_3d.conv_input[0].weight.detach().clone()
    assert torch.equal(t0, s0)
    for _ in range(2):
        loss = wl.step()
        assert torch.isfinite(loss)
    log = wl.last_log
    for k in ('sup.sup_3d.loss', 'ssl.unlab.hard_pseudo_3d.loss', 'ssl.unlab.metrics.tea', 'ssl.weight',
              'ssl.ema_decay', 'loss'):
        assert k in log, (k, list(log))
    s1 = wl.model.student.detector_3d.model.backbone_3d.conv_input[0].weight.detach()
    t1 = wl.model.teacher.detector_3d.model.backbone_3d.conv_input[0].weight.detach()
    assert not torch.equal(s1, s0)
    # teacher after 2 iterations: EMA ran at the start of iteration 2 with d = 1 - 1/100 on the
    # student of iteration 1 (iteration 1's EMA saw identical weights)
    assert not torch.equal(t1, t0) and (t1 - t0).abs().max() < (s1 - s0).abs().max()
    assert wl.model.iter == 1 and not wl.model.teacher.training


def test_detmatch_full_iteration(dev):
    """Full DetMatch recipe (2D + 3D, Hungarian matching, consistency): one iteration."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, dev)
    loss = wl.step()
    assert torch.isfinite(loss)
    log = wl.last_log
    want = ['sup.sup_3d.loss', 'sup.stu.loss_rpn_cls', 'sup.stu.loss_rpn_bbox', 'sup.stu.loss_cls',
            'sup.stu.acc', 'sup.stu.loss_bbox', 'ssl.unlab.hard_pseudo_3d.loss',
            'ssl.unlab.hard_pseudo_2d.loss_rpn_cls', 'ssl.unlab.hard_pseudo_2d.loss_cls',
            'ssl.unlab.hard_pseudo_2d.acc', 'ssl.unlab.2D_to_3D_hung.cls_loss',
            'ssl.unlab.2D_to_3D_hung.l1_loss', 'ssl.unlab.2D_to_3D_hung.iou_loss',
            'ssl.unlab.metrics.num_tea_hung', 'ssl.unlab.metrics.2D_to_3D_hung']
    for k in want:
        assert k in log and torch.isfinite(log[k]), (k, sorted(log))
    assert 'ssl.unlab.hard_pseudo_2d.loss_bbox' not in log      # detached by the recipe
    loss2 = wl.step()
    assert torch.isfinite(loss2)


def test_fused_optimizer_matches_torch(dev):
    """dm_adamw_step_f32 / dm_sgd_step_f32 over the flat arenas (with the clip coefficient folded in)
    == torch.optim.AdamW / SGD after clip_grad_norm_, over several steps."""
    import copy
    import torch.nn as nn
    from detmatch_amd.mm3d import runner as R
    from detmatch_amd.mm3d.parallel import FlatGradDDP

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.student = nn.ModuleDict(dict(
                detector_3d=nn.Sequential(nn.Linear(37, 53), nn.ReLU(), nn.Linear(53, 7)),
                detector_2d=nn.Sequential(nn.Linear(37, 29), nn.ReLU(), nn.Linear(29, 3))))

        def forward(self, x):
            return self.student['detector_3d'](x).square().mean() + self.student['detector_2d'](x).abs().mean()

    cfg = {'constructor': 'HybridOptimizerConstructor',
           'student.detector_3d': dict(type='AdamW', lr=0.01, betas=(0.95, 0.99), weight_decay=0.01),
           'student.detector_2d': dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=0.0001)}
    torch.manual_seed(0)
    a = Toy().to(dev)
    b = copy.deepcopy(a)
    ddp = FlatGradDDP(a, broadcast=False)
    ddp.build_param_arena()
    opt_a = R.build_optimizer(a, cfg)
    assert opt_a.enable_fused(ddp) == 2
    opt_b = R.build_optimizer(b, cfg)
    for it in range(5):
        x = torch.randn(16, 37, device=dev) * (30.0 if it % 2 else 1.0)   # clip active on odd steps
        ddp.zero_grad()
        a(x).backward()
        ddp.finish()
        norm_a, opt_a.grad_scale = ddp.clip_coef(max_norm=1.0)
        opt_a.step()
        opt_b.zero_grad()
        b(x).backward()
        norm_b = torch.nn.utils.clip_grad_norm_(list(b.parameters()), 1.0)
        opt_b.step()
        assert torch.allclose(norm_a, norm_b, rtol=1e-5)
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-6), (it, n)
    sd = opt_a.state_dict()     # torch-compatible state view
    assert len(sd['state'][0]) == 4 and 'exp_avg' in sd['state'][0][0] and 'momentum_buffer' in sd['state'][1][0]
    # resume parity (hybrid_optimizer.py:41-68): a fused optimizer restored from (i) its own and
    # (ii) a plain torch HybridOptimizer checkpoint continues exactly like the one that never stopped
    resumed = []
    for ckpt, fuse_first in ((copy.deepcopy(sd), True), (copy.deepcopy(opt_b.state_dict()), False)):
        c = copy.deepcopy(b)
        ddp_c = FlatGradDDP(c, broadcast=False)
        ddp_c.build_param_arena()
        opt_c = R.build_optimizer(c, cfg)
        if fuse_first:
            assert opt_c.enable_fused(ddp_c) == 2
        opt_c.load_state_dict(ckpt)
        if not fuse_first:
            assert opt_c.enable_fused(ddp_c) == 2
        assert opt_c.num_step_updated == 5
        resumed.append((c, ddp_c, opt_c))
    for it in range(2):
        x = torch.randn(16, 37, device=dev)
        opt_b.zero_grad()
        b(x).backward()
        torch.nn.utils.clip_grad_norm_(list(b.parameters()), 1.0)
        opt_b.step()
        for c, ddp_c, opt_c in resumed:
            ddp_c.zero_grad()
            c(x).backward()
            ddp_c.finish()
            _, opt_c.grad_scale = ddp_c.clip_coef(max_norm=1.0)
            opt_c.step()
            for (n, pc), pb in zip(c.named_parameters(), b.parameters()):
                assert torch.allclose(pc, pb, rtol=2e-5, atol=2e-6), (it, n)


@pytest.mark.parametrize('ssl_cfg', ['confthr_pvrcnn', None])
def test_early_backward_and_prefetch_do_not_change_gradients(dev, ssl_cfg):
    """Scheduling options (self-contained passes back-propagated as soon as their losses exist;
    geometry of all passes issued up front) leave the accumulated gradient and the logged losses
    unchanged (None = the full DetMatch recipe)."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    flats, logs = [], []
    for early in (True, False):
        wl = DetMatchTrainWorkload(2, dev, ssl_cfg=ssl_cfg)
        wl.model.early_backward = early
        wl.model.two_lanes, wl.model.lane_mode = False, None      # one stream: the per-pass early backward only exists there
        if not early:   # also disable the geometry prepass for the plain run
            for m in wl.model.lab_ssl_modules + wl.model.unlab_ssl_modules:
                if hasattr(m, 'prefetch'):
                    m.prefetch = lambda ssl_obj, d: None
        torch.manual_seed(123)
        wl.step()
        flats.append(wl.ddp.flat.clone())
        logs.append({k: float(v) for k, v in wl.last_log.items()})
        del wl
    a, b = flats
    assert torch.isfinite(a).all() and a.abs().sum() > 0
    rel = (a - b).norm() / b.norm()
    assert rel < 1e-3, float(rel)      # float atomics: run-to-run noise ~1e-5..1e-4
    assert set(logs[0]) == set(logs[1])
    for k in logs[0]:
        assert abs(logs[0][k] - logs[1][k]) <= 1e-3 * max(1.0, abs(logs[1][k])), k


@pytest.mark.parametrize('mode', ['branches', 'glue'])
def test_lanes_do_not_change_gradients(dev, mode):
    """Full DetMatch recipe on several HIP streams (data-flow edges turned into event waits) gives the
    same accumulated gradient and losses as the serial order.  'branches': student-3D / 2D detectors /
    teacher-3D + glue lanes; 'glue': only the pseudo-label glue on a side stream, teacher inference
    issued ahead of the supervised passes."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    for lanes in (mode, None):
        wl = DetMatchTrainWorkload(2, dev)
        wl.model.two_lanes, wl.model.lane_mode = False, lanes
        # the one-pass 2D trunk is a property of the one-stream orders only (its own test is
        # test_shared_student_2d_trunk_equals_separate_passes): off on both sides here
        wl.model.share_2d_trunk = False
        torch.manual_seed(321)       # one iteration: no feedback through updated weights
        wl.step()
        torch.cuda.synchronize()
        out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
        del wl
    (ga, la), (gb, lb) = out
    assert torch.isfinite(ga).all() and ga.abs().sum() > 0
    assert set(la) == set(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 2e-3, float(rel)


def test_early_2d_backward_does_not_change_gradients(dev, monkeypatch):
    """'branches' with the shared 2D trunk: the unlabeled 2D module's losses and the deferred trunk backward issued on
    the 2D lane right after that module (SSL._early_2d_backward) give the same accumulated gradients and losses as
    back-propagating everything at the end of the iteration; likewise the early issue of the unlabeled passes."""
    from detmatch_amd.mm3d import ssl
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    for early in (True, False):
        monkeypatch.setattr(ssl, '_EARLY_2D_BWD', early)
        monkeypatch.setattr(ssl, '_ISSUE_EARLY', early)
        monkeypatch.setenv('DM_TWO_LANES', '1')      # the bench default
        wl = DetMatchTrainWorkload(2, dev)
        assert wl.model.two_lanes and wl.model.share_2d_trunk
        torch.manual_seed(321)       # one iteration: no feedback through updated weights
        wl.step()
        torch.cuda.synchronize()
        out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
        del wl
    (ga, la), (gb, lb) = out
    assert torch.isfinite(ga).all() and ga.abs().sum() > 0
    assert set(la) == set(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 2e-3, float(rel)


def test_confthr_frcnn_iteration(dev):
    """2D-only SSL recipe (configs/detmatch/001/confthr_frcnn): teacher Faster R-CNN ->
    NMS(0.7) -> un-augment / re-augment -> hard pseudo labels for the student; two iterations."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, dev, ssl_cfg='confthr_frcnn')
    w3 = wl.model.student.detector_3d.model.dense_head.conv_cls.weight.detach().clone()
    w2 = wl.model.student.detector_2d.rpn_head.rpn_cls.weight.detach().clone()
    for _ in range(2):
        assert torch.isfinite(wl.step())
    log = wl.last_log
    for k in ('sup.stu.loss_rpn_cls', 'sup.stu.loss_rpn_bbox', 'sup.stu.loss_cls', 'sup.stu.loss_bbox',
              'ssl.unlab.hard_pseudo_2d.loss_rpn_cls', 'ssl.unlab.hard_pseudo_2d.loss_cls',
              'ssl.unlab.metrics.num_tea', 'loss'):
        assert k in log and torch.isfinite(log[k]), (k, sorted(log))
    assert 'ssl.unlab.hard_pseudo_2d.loss_bbox' not in log and 'sup.sup_3d.loss' not in log
    assert not torch.equal(w2, wl.model.student.detector_2d.rpn_head.rpn_cls.weight)
    # the 3D student gets no gradient in this recipe: never-used parameters are skipped by the fused
    # optimizer kernels (liveness mask), exactly as torch.optim skips `.grad is None` — not even weight decay
    assert torch.equal(wl.model.student.detector_3d.model.dense_head.conv_cls.weight, w3)


@pytest.mark.parametrize('recipe', ['pretrain_pvrcnn', 'pretrain_frcnn'])
def test_pretrain_recipes_run_through_epoch_based_runner(dev, recipe):
    """configs/detmatch/001/pretrain_{pvrcnn,frcnn}: EpochBasedRunner + the stand-alone detector
    train_step, 2 epochs x 3 iterations on one synthetic batch: the loss goes down."""
    from detmatch_amd.pcdet.workload import PretrainWorkload
    wl = PretrainWorkload(2, dev, recipe=recipe, iters_per_epoch=3, max_epochs=2)
    wl.run()
    r = wl.runner
    assert r.iter == 6 and r.epoch == 2
    losses = [float(v) for v in r.log_buffer['loss']]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    if recipe == 'pretrain_pvrcnn':
        assert len(r.log_buffer['grad_norm']) == 6
        g = r.optimizer.param_groups[0]
        assert g['lr'] != g['initial_lr'] and g['betas'][0] != 0.9      # cyclic schedules are live


def test_baseline_config0_pretrain_pvrcnn_one_iteration_batch_one(dev):
    """BASELINE.json configs[0]: configs/detmatch/001/pretrain_pvrcnn/split_0.py, ONE iteration at batch size 1 —
    the reference's CPU-runnable plumbing case (voxelise + sparse backbone + losses).  The product has no CPU path
    by design, so the same workload runs on the device: the loss and the clipped gradient norm are finite, the
    gradient reaches every head and the first sparse layer, one optimizer step moves the weights, and the voxel
    grid of the single frame equals the oracle's (the CPU restatement of the reference's voxeliser)."""
    import oracle
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import PretrainWorkload
    wl = PretrainWorkload(1, dev, recipe='pretrain_pvrcnn', iters_per_epoch=1, max_epochs=1)
    m = wl.model.model
    first = m.backbone_3d.conv_input[0].weight
    before = first.detach().clone()
    wl.run()
    r = wl.runner
    assert r.iter == 1
    log = {k: float(v[-1]) for k, v in r.log_buffer.items()}
    assert np.isfinite(log['loss']) and log['loss'] > 0
    assert np.isfinite(log['grad_norm']) and log['grad_norm'] > 0
    for head in (m.dense_head, m.point_head, m.roi_head):      # every head took part: its parameters moved
        assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in head.parameters()), type(head).__name__
    assert not torch.equal(first.detach(), before)
    # the frame's voxels against the oracle
    pts = wl.loader[0]['points'][0]
    ov, oc, on = oracle.hard_voxelize(pts.cpu().numpy(), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    from detmatch_amd import voxel
    v, c, n, mean, counts = voxel.voxelize_batch([pts], synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
    assert np.array_equal(c[:, 1:].cpu().numpy(), oc) and np.array_equal(n.cpu().numpy(), on)


def test_filter_by_nms_3d_multiclass(dev):
    """ssl_modules/bbox_utils.py:203-279 + core/post_processing/box3d_nms.py: per-class rotated BEV NMS;
    the kept set per class equals the oracle's greedy NMS on the same rotated rectangles, survivors keep
    their full score vectors and the score threshold / max_num are honoured."""
    from detmatch_amd.mm3d.bbox_utils import filter_by_nms
    from detmatch_amd.mm3d.box3d import LiDARInstance3DBoxes
    rng = np.random.default_rng(0)
    n = 300
    base = np.stack([rng.uniform(5, 40, 40), rng.uniform(-15, 15, 40), rng.uniform(-1.8, -1.2, 40),
                     rng.uniform(1.4, 2.0, 40), rng.uniform(3.2, 4.6, 40), rng.uniform(1.4, 1.8, 40),
                     rng.uniform(-3.1, 3.1, 40)], 1)
    b = (base[rng.integers(0, 40, n)] + rng.normal(0, 0.25, (n, 7))).astype(np.float32)
    scores = rng.uniform(0, 1, (n, 3)).astype(np.float32) ** 2
    boxes = LiDARInstance3DBoxes(torch.from_numpy(b).to(dev))
    cfg = dict(nms_pre=-1, score_thr=0.3, max_num=60, use_rotate_nms=True, nms_thr=0.3)
    (kept, kept_scores, kept_labels), = filter_by_nms([(boxes, torch.from_numpy(scores).to(dev))], cfg, True,
                                                      return_labels=True)
    assert len(kept) == len(kept_scores) == len(kept_labels) <= 60 and kept_scores.shape[1] == 3
    # independent geometry: corners by mmdet3d's iou3d convention (clockwise by r), polygon clipping
    from _polyclip import corners_clockwise, intersection_area
    bev = boxes.bev.cpu().numpy().astype(np.float64)    # (cx, cy, w, h, r)
    polys = [corners_clockwise(*row) for row in bev]
    area = bev[:, 2] * bev[:, 3]
    want = []
    for c in range(3):
        idx = np.nonzero(scores[:, c] > 0.3)[0]
        order = idx[np.argsort(-scores[idx, c], kind='stable')]
        alive = np.ones(len(order), bool)
        for p in range(len(order)):
            if not alive[p]:
                continue
            want.append((float(scores[order[p], c]), c, int(order[p])))
            for q in range(p + 1, len(order)):
                if alive[q]:
                    it = intersection_area(polys[order[p]], polys[order[q]])
                    if it / max(area[order[p]] + area[order[q]] - it, 1e-6) > 0.3:
                        alive[q] = False
    want.sort(key=lambda t: -t[0])
    want = want[:60]
    got_rows = {(int(l), tuple(np.round(x, 4))) for l, x in zip(kept_labels.cpu().numpy(), kept.tensor.cpu().numpy())}
    want_rows = {(c, tuple(np.round(b[i], 4))) for _, c, i in want}
    assert got_rows == want_rows
    for l, s, x in zip(kept_labels.cpu().numpy(), kept_scores.cpu().numpy(), kept.tensor.cpu().numpy()):
        i = int(np.argmin(np.abs(b - x).sum(1)))
        assert np.allclose(s, scores[i]) and s[l] > 0.3          # the FULL score vector of the survivor


def test_fused_optimizer_skips_never_used_parameters(dev):
    """A branch that never receives a gradient (confthr_pvrcnn's 2D student) is left untouched by
    the fused kernels (liveness mask), exactly like torch optimizers skip `.grad is None`; a
    parameter used once keeps being stepped (weight decay, momentum) with zero gradients."""
    import copy
    import torch.nn as nn
    from detmatch_amd.mm3d import runner as R
    from detmatch_amd.mm3d.parallel import FlatGradDDP

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.student = nn.ModuleDict(dict(
                detector_3d=nn.ModuleDict(dict(a=nn.Linear(37, 53), once=nn.Linear(53, 53), b=nn.Linear(53, 7))),
                detector_2d=nn.Sequential(nn.Linear(37, 29), nn.ReLU(), nn.Linear(29, 3))))

        def forward(self, x, use_once):
            d = self.student['detector_3d']
            h = torch.relu(d['a'](x))
            if use_once:
                h = h + d['once'](h)
            return d['b'](h).square().mean()               # detector_2d: never used

    cfg = {'constructor': 'HybridOptimizerConstructor',
           'student.detector_3d': dict(type='AdamW', lr=0.01, betas=(0.95, 0.99), weight_decay=0.01),
           'student.detector_2d': dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=0.01)}
    torch.manual_seed(0)
    a = Toy().to(dev)
    b = copy.deepcopy(a)
    ddp = FlatGradDDP(a, broadcast=False)
    ddp.build_param_arena()
    opt_a = R.build_optimizer(a, cfg)
    assert opt_a.enable_fused(ddp) == 2
    opt_b = R.build_optimizer(b, cfg)
    w2d = [p.detach().clone() for p in a.student['detector_2d'].parameters()]
    for it in range(4):
        x = torch.randn(16, 37, device=dev)
        ddp.zero_grad()
        a(x, it == 0).backward()
        ddp.finish()
        opt_a.step()
        opt_b.zero_grad()            # torch default set_to_none... mmcv semantics: keep tensors
        for p in b.parameters():
            if p.grad is not None:
                p.grad.zero_()
        b(x, it == 0).backward()
        opt_b.step()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-6), (it, n)
    for p, w in zip(a.student['detector_2d'].parameters(), w2d):
        assert torch.equal(p, w)
    sd = opt_a.state_dict()
    assert len(sd['state'][1]) == 0                         # no state for the dead branch
    steps = [st['step'] for st in sd['state'][0].values()]
    assert all(int(s) == 4 for s in steps) and len({s.data_ptr() for s in steps}) == len(steps)


def test_fused_optimizer_counts_steps_per_parameter(dev):
    """torch.optim keeps its state per parameter: a parameter whose FIRST gradient arrives at step 3 starts
    its AdamW bias correction / its SGD momentum buffer there.  The fused kernels take the
    per-block first-gradient step (dm_adamw_step_blocks_f32 / dm_sgd_step_blocks_f32): the late branch
    follows torch.optim from its first update on, and the checkpointed `step` is the parameter's own."""
    import copy
    import torch.nn as nn
    from detmatch_amd.mm3d import runner as R
    from detmatch_amd.mm3d.parallel import FlatGradDDP

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.student = nn.ModuleDict(dict(
                detector_3d=nn.ModuleDict(dict(a=nn.Linear(37, 53), late=nn.Linear(53, 53), b=nn.Linear(53, 7))),
                detector_2d=nn.ModuleDict(dict(a=nn.Linear(37, 29), late=nn.Linear(29, 29), b=nn.Linear(29, 3)))))

        def forward(self, x, use_late):
            out = 0
            for d in (self.student['detector_3d'], self.student['detector_2d']):
                h = torch.relu(d['a'](x))
                if use_late:
                    h = h + d['late'](h)
                out = out + d['b'](h).square().mean()
            return out

    cfg = {'constructor': 'HybridOptimizerConstructor',
           'student.detector_3d': dict(type='AdamW', lr=0.01, betas=(0.9, 0.99), weight_decay=0.01),
           'student.detector_2d': dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=0.01)}
    torch.manual_seed(0)
    a = Toy().to(dev)
    b = copy.deepcopy(a)
    ddp = FlatGradDDP(a, broadcast=False)
    ddp.build_param_arena()
    opt_a = R.build_optimizer(a, cfg)
    assert opt_a.enable_fused(ddp) == 2
    opt_b = R.build_optimizer(b, cfg)
    for it in range(7):
        x = torch.randn(16, 37, device=dev)
        ddp.zero_grad()
        a(x, it >= 2).backward()
        ddp.finish()
        opt_a.step()
        for p in b.parameters():            # mmcv zero_grad semantics: tensors stay, never-used stay None
            if p.grad is not None:
                p.grad.zero_()
        b(x, it >= 2).backward()
        opt_b.step()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=3e-5, atol=3e-6), (it, n)
    sd = opt_a.state_dict()
    names = [n for n, _ in a.student['detector_3d'].named_parameters()]
    steps = {n: int(st['step']) for n, st in zip(names, sd['state'][0].values())}
    assert steps['late.weight'] == 5 and steps['a.weight'] == 7, steps
    # every parameter is live now and one started late: the first-step vector stays in use
    assert opt_a._fused[0].first is not None and int(opt_a._fused[0].first.max()) == 3
    # a checkpoint of this state continues identically under plain torch optimizers and under fused ones
    c = copy.deepcopy(b)
    opt_c = R.build_optimizer(c, cfg)
    opt_c.load_state_dict(copy.deepcopy(sd))
    d = copy.deepcopy(b)
    ddp_d = FlatGradDDP(d, broadcast=False)
    ddp_d.build_param_arena()
    opt_d = R.build_optimizer(d, cfg)
    opt_d.load_state_dict(copy.deepcopy(sd))
    assert opt_d.enable_fused(ddp_d) == 2
    x = torch.randn(16, 37, device=dev)
    for m, o, dd in ((a, opt_a, ddp), (c, opt_c, None), (d, opt_d, ddp_d)):
        if dd is not None:
            dd.zero_grad()
        else:
            o.zero_grad()
        m(x, True).backward()
        if dd is not None:
            dd.finish()
        o.step()
    for (n, pa), pc, pd in zip(a.named_parameters(), c.parameters(), d.parameters()):
        assert torch.allclose(pa, pc, rtol=3e-5, atol=3e-6), n
        assert torch.allclose(pa, pd, rtol=3e-5, atol=3e-6), n


def test_gradient_exchange_code_path_on_one_rank_nccl(dev, monkeypatch):
    """`exchange='rs_ag'` (reduce_scatter_tensor + all_gather_into_tensor per bucket, RCCL) and
    `mode='hooks'` (buckets issued from autograd hooks, strictly in index order, overlapping backward) need
    the nccl backend, so the gloo tests cannot run them.  A ONE-rank nccl group on the one GPU executes the
    very same calls (identities on one rank): (1) on a deterministic network the exchanged gradients equal
    the default path's bit for bit; (2) a full DetMatch iteration runs through it and agrees with the default
    path within the run-to-run spread of its atomics."""
    import socket
    import torch.distributed as dist
    import torch.nn as nn
    from detmatch_amd.mm3d.parallel import FlatGradDDP
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        net = nn.Sequential(nn.Linear(64, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 8)).to(dev)
        x = torch.randn(32, 64, device=dev)
        grads = []
        for kw in (dict(), dict(mode='hooks', exchange='rs_ag', always_exchange=True),
                   dict(mode='collect', exchange='all_reduce', always_exchange=True)):
            ddp = FlatGradDDP(net, bucket_bytes=64 << 10, broadcast=False, **kw)
            assert len(ddp.buckets) >= 3
            for _ in range(2):
                ddp.zero_grad()
                ddp(x).square().mean().backward()
                ddp.finish()
            grads.append(ddp.flat.clone())
            if kw:
                assert ddp.n_collectives == 2 * len(ddp.buckets) * (2 if kw['exchange'] == 'rs_ag' else 1)
            for p in net.parameters():
                p.grad = None
        assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
        assert float(grads[0].abs().sum()) > 0
        # a full DetMatch iteration through hooks + rs_ag
        losses = []
        for special in (False, True):
            monkeypatch.setenv('DM_GRAD_MODE', 'hooks' if special else 'collect')
            monkeypatch.setenv('DM_GRAD_EXCHANGE', 'rs_ag' if special else 'all_reduce')
            wl = DetMatchTrainWorkload(2, dev, seed=0)
            wl.ddp.always_exchange = special
            assert wl.ddp.mode == ('hooks' if special else 'collect') and wl.ddp.exchange == ('rs_ag' if special else 'all_reduce')
            ls = [float(wl.step().detach()) for _ in range(3)]
            torch.cuda.synchronize()
            if special:
                assert wl.ddp.n_collectives == 3 * 2 * len(wl.ddp.buckets)
            losses.append(ls)
            del wl
        # the first two iterations agree to rounding; by the third the random-init model has dropped its loss
        # 14x and differences of that size have been amplified through tie-prone proposals and matches
        np.testing.assert_allclose(losses[0][:2], losses[1][:2], rtol=2e-3)
        np.testing.assert_allclose(losses[0][2], losses[1][2], rtol=0.25)
    finally:
        dist.destroy_process_group()


def test_shared_student_2d_trunk_equals_separate_passes(dev, monkeypatch):
    """The student's backbone + FPN + RPN convolutions run ONCE per iteration on the labeled and the unlabeled
    images together (FasterRCNN.prefetch_trunk, frozen BatchNorm makes the samples independent) with the trunk
    backward deferred until both heads' gradients are in: same losses and the same parameter update as two
    half-batch passes, up to the summation order of the weight gradients."""
    from detmatch_amd.mm2d.faster_rcnn import FasterRCNN
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    res = []
    for share in ('1', '0'):
        monkeypatch.setenv('DM_SHARE_2D_TRUNK', share)
        monkeypatch.setenv('DM_LOOKAHEAD', '0')
        calls = []
        orig = FasterRCNN._trunk          # backbone + FPN + RPN convolutions (one chained call, or extract_feat + rpn_head)
        monkeypatch.setattr(FasterRCNN, '_trunk', lambda self, img, _o=orig: (calls.append((id(self), int(img.shape[0]))), _o(self, img))[1])
        wl = DetMatchTrainWorkload(2, dev, seed=3)
        stu = wl.model.student.detector_2d
        w0 = stu.neck.lateral_convs[0].conv.weight.detach().clone()
        logs = []
        for _ in range(2):
            wl.step()
            logs.append({k: float(v) for k, v in wl.last_log.items() if 'loss' in k})
        torch.cuda.synchronize()
        mine = [b for i, b in calls if i == id(stu)]
        assert mine == ([4, 4] if share == '1' else [2, 2, 2, 2]), mine
        res.append((logs, (stu.neck.lateral_convs[0].conv.weight.detach() - w0).clone(),
                    stu.backbone.layer3[0].conv2.weight.detach().clone()))
        monkeypatch.setattr(FasterRCNN, '_trunk', orig)
        del wl
    (la, da, wa), (lb, db, wb) = res
    # First iteration, same weights.  The trunk outputs agree up to the summation order of the split-K layers
    # (their split depends on the batch size): the RPN losses (sums over the sampled anchors) agree to fp32
    # accuracy; the RoI head sees the top-scoring proposals of a RANDOM-INIT RPN, whose scores are all within
    # 1e-3 of each other, so a handful of its 1024 RoIs differ and its classification loss moves by ~5e-4.
    for k in la[0]:
        tol = 2e-3 if ('loss_cls' in k or 'loss_bbox' in k or k == 'loss') else 1e-5
        assert la[0][k] == pytest.approx(lb[0][k], rel=tol, abs=1e-6), k
    # (+ 1e-8: two iterations at the warm-up learning rate move these weights by ~1e-7, a few dozen fp32 ulps of the
    # weights themselves, so the difference of two updates is quantised in steps of 3.7e-9)
    assert float((da - db).abs().max()) <= 2e-2 * float(db.abs().max()) + 1e-8
    assert float((wa - wb).abs().max()) <= 1e-3 * float(wb.abs().max())
    for k in ('sup.stu.loss_rpn_cls', 'sup.stu.loss_cls', 'loss'):
        assert la[1][k] == pytest.approx(lb[1][k], rel=1e-2), k                    # second: after one update


def test_lookahead_geometry_is_scheduling_only(dev, monkeypatch):
    """IterBasedSSLRunner's look-ahead (batches of iteration i + 1 drawn before iteration i is issued, their
    weight-independent 3D geometry prepared on a side stream under iteration i) changes WHEN things are issued,
    not what is computed: the same batches in the same order, and the same losses as the plain order — the
    first iteration to rounding of its atomics, the following ones within the spread those leave after an update."""
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    logs = {}
    for ahead in ('1', '0'):
        wl = DetMatchTrainWorkload(2, dev, seed=11)
        wl.runner.lookahead = ahead == '1'        # (the class default comes from DM_LOOKAHEAD at import)
        out = []
        for _ in range(3):
            wl.step()
            out.append({k: float(v) for k, v in wl.last_log.items() if 'loss' in k})
        torch.cuda.synchronize()
        logs[ahead] = out
        del wl
    for k in logs['1'][0]:
        assert logs['1'][0][k] == pytest.approx(logs['0'][0][k], rel=1e-5, abs=1e-6), k
    for it in (1, 2):
        assert logs['1'][it]['loss'] == pytest.approx(logs['0'][it]['loss'], rel=2e-2), it


@pytest.mark.timeout(300)
def test_shipped_order_soak_300_iterations(dev):
    """The configuration bench.py measures (three lanes, chains, early issue, shared 2D trunk, batched FPS, lazy glue),
    300 iterations in a row: finishes (round 5's reproducer of the lane dead-lock wedged within ~100-400), stays finite,
    the FC stacks ran on the library's own GEMMs and no vendor GEMM ever needed a cross-lane edge."""
    import time
    from detmatch_amd import _lib
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    wl = DetMatchTrainWorkload(2, dev)
    assert wl.model.two_lanes and wl.model.share_2d_trunk
    for h in wl.runner._hooks:             # (a random-init model diverges over hundreds of steps at the recipe's rate)
        if getattr(h, 'base_lr', None):
            h.base_lr = [lr * 1e-4 for lr in h.base_lr]
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    edges, own = _lib.BLAS_TURNS[1], _lib.OWN_LINEAR_CALLS[0] + _lib.FC_GEMM_CALLS[0]
    t0 = time.time()
    for i in range(300):
        loss = wl.step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert torch.isfinite(loss) and dt < 120, dt
    assert _lib.OWN_LINEAR_CALLS[0] + _lib.FC_GEMM_CALLS[0] - own >= 300 * 60
    assert _lib.BLAS_TURNS[1] == edges, 'vendor GEMMs were issued on two lanes'


def test_side_stream_weight_gradients_do_not_change_gradients(dev):
    """chain.SIDE_WGRAD (the weight-gradient halves of the chained dense / set-abstraction backward passes on the side
    stream, read by FlatGradDDP.collect behind their events): same accumulated gradient, same losses as with every
    backward kernel on the main lane."""
    from detmatch_amd import chain
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    for side in (True, False):
        wl = DetMatchTrainWorkload(2, dev)
        wl.model.side_wgrad = side
        assert wl.model.two_lanes and wl.ddp.mode == 'collect'
        torch.manual_seed(321)
        wl.step()
        torch.cuda.synchronize()
        assert not chain.SIDE_WGRAD[0]               # only for the length of an iteration
        out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
        del wl
    (ga, la), (gb, lb) = out
    assert torch.isfinite(ga).all() and ga.abs().sum() > 0
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 2e-3, float(rel)


def test_key_point_encoder_on_the_side_stream_does_not_change_gradients(dev):
    """pcdet/detector.py:PFE_SIDE (the key-point encoder beside the BEV backbone, forward and backward): same accumulated
    gradient and losses as the module list's own order."""
    from detmatch_amd.pcdet import detector
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    saved = detector.PFE_SIDE[0]
    try:
        for side in (True, False):
            detector.PFE_SIDE[0] = side
            wl = DetMatchTrainWorkload(2, dev)
            torch.manual_seed(321)
            wl.step()
            torch.cuda.synchronize()
            out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
            del wl
    finally:
        detector.PFE_SIDE[0] = saved
    (ga, la), (gb, lb) = out
    assert torch.isfinite(ga).all() and ga.abs().sum() > 0
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 2e-3, float(rel)


def _zero_lr(wl):
    for h in wl.runner._hooks:
        if getattr(h, 'base_lr', None):
            h.base_lr = [0.0 for _ in h.base_lr]


def test_teacher_ahead_of_the_previous_backward_is_scheduling_only(dev):
    """ssl._TEACHER_AHEAD: from the second iteration on, the geometry of all passes and the teacher's 2D pass wait for the
    previous iteration's EMA and the batch, not for its last backward / optimizer step — they run underneath that tail.
    With the learning rate at zero the weights stay put, so iterations 2 and 3 must give the same losses with and without
    (to the rounding of the atomics): anything the early work read too early, or wrote under a reader, would show."""
    from detmatch_amd.mm3d import ssl
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    logs = {}
    saved = ssl._TEACHER_AHEAD
    try:
        for ahead in (True, False):
            ssl._TEACHER_AHEAD = ahead
            wl = DetMatchTrainWorkload(2, dev, seed=5)
            assert wl.runner.draw_ahead and wl.model.two_lanes
            _zero_lr(wl)
            out = []
            for _ in range(4):
                wl.step()
                out.append({k: float(v) for k, v in wl.last_log.items()})
            torch.cuda.synchronize()
            assert (getattr(wl.model, '_ema_done', None) is not None)
            logs[ahead] = out
            del wl
    finally:
        ssl._TEACHER_AHEAD = saved
    for it in range(4):
        for k, v in logs[True][it].items():
            assert v == pytest.approx(logs[False][it][k], rel=2e-4, abs=1e-6), (it, k)


def test_supervised_2d_backward_on_its_lane_does_not_change_gradients(dev):
    """ssl._SUP_BWD_PER_LANE (off by default: measured neutral): the supervised 2D losses back-propagated on the 2D lane
    right behind their forward, the 3D ones on the main lane without waiting for the 2D lane — same accumulated gradient."""
    from detmatch_amd.mm3d import ssl
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload
    out = []
    saved = ssl._SUP_BWD_PER_LANE
    try:
        for per_lane in (True, False):
            ssl._SUP_BWD_PER_LANE = per_lane
            wl = DetMatchTrainWorkload(2, dev)
            torch.manual_seed(321)
            wl.step()
            torch.cuda.synchronize()
            out.append((wl.ddp.flat.clone(), {k: float(v) for k, v in wl.last_log.items()}))
            del wl
    finally:
        ssl._SUP_BWD_PER_LANE = saved
    (ga, la), (gb, lb) = out
    assert torch.isfinite(ga).all() and ga.abs().sum() > 0
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    rel = (ga - gb).norm() / gb.norm()
    assert rel < 2e-3, float(rel)
