"""CPU tests of the teacher-student host layer: Hungarian matching against
the reference-generated fixture, SSL scalar schedules, loss bookkeeping, SSL modules that are pure
tensor glue, the hybrid optimizer, the LR warm-up and the iteration runner."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from detmatch_amd.mm3d import losses as L
from detmatch_amd.mm3d import runner as R
from detmatch_amd.mm3d import ssl_modules as M
from detmatch_amd.mm3d.registry import build_assigner
from detmatch_amd.mm3d.ssl import SSL

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'hungarian.npz')

ASSIGNER_CFG = dict(type='ModHungarianAssigner',
                    cls_cost=dict(type='DoubleSidedFocalLossCost', weight=2.0),
                    reg_cost=dict(type='BBoxL1Cost', weight=5.0),
                    iou_cost=dict(type='IoUCost', iou_mode='giou', weight=2.0))


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLDEN)


def test_lap_matches_scipy_fixture(gold):
    """dm_lap_host == scipy.optimize.linear_sum_assignment on the stored cost matrices: the
    assignment is bit-exact (same (row, col) pairs)."""
    names = ['l%d' % i for i in range(int(gold['n_lap']))] + \
            ['c%d' % i for i in range(int(gold['n_cases'])) if ('c%d_cost' % i) in gold]
    for n in names:
        rows, cols = L.linear_sum_assignment(torch.from_numpy(gold[n + '_cost']))
        assert np.array_equal(rows, gold[n + '_rows']), n
        assert np.array_equal(cols, gold[n + '_cols']), n


def test_mod_hungarian_assigner_matches_reference(gold):
    assigner = build_assigner(ASSIGNER_CFG)
    meta = dict(img_shape=(375, 1242, 3))
    for ci in range(int(gold['n_cases'])):
        g = lambda k: torch.from_numpy(gold['c%d_%s' % (ci, k)])
        res = assigner.assign(g('pred_norm'), g('pred_logits'), g('gt'), g('gt_logits'), meta)
        assert np.array_equal(res.gt_inds.numpy(), gold['c%d_gt_inds' % ci]), ci
        if bool(gold['c%d_has_cost' % ci]):
            np.testing.assert_allclose(res.max_overlaps.numpy(), gold['c%d_max_overlaps' % ci],
                                       rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(res.iou_cost.numpy(), gold['c%d_iou_cost' % ci],
                                       rtol=1e-5, atol=1e-6)
        else:
            assert res.max_overlaps is None


def test_lap_nan_is_an_error():
    c = torch.tensor([[1.0, float('nan')], [0.5, 2.0]])
    with pytest.raises(ValueError):
        L.linear_sum_assignment(c)


# ---------------------------------------------------------------- SSL scalar schedules
class _Dummy(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.w = nn.Parameter(torch.ones(3))


def _bare_ssl(ema=None, weight=None):
    s = SSL.__new__(SSL)
    nn.Module.__init__(s)
    s.ema_params = ema or dict(ema_decay=0.999, true_avg_rampup=True, rampup_start_decay=0.99)
    s.ema_decay = s.ema_params['ema_decay']
    s.rampup_start_decay = s.ema_params.get('rampup_start_decay', 0.5)
    wp = weight or dict(weight=1)
    s.ssl_weight = wp['weight']
    s.ssl_weight_rampup_start_iter = wp.get('weight_rampup_start_iter', 0)
    s.ssl_weight_rampup_num_iter = wp.get('weight_rampup_num_iter', 0)
    s.iter = 0
    return s


def test_ema_decay_sequence():
    """ssl.py:129-144 with the DetMatch config: d_t = min(1 - 1/(t + 100), 0.999)."""
    s = _bare_ssl()
    for it in (0, 1, 10, 899, 900, 901, 5000):
        s.iter = it
        assert s._get_curr_ema_decay() == pytest.approx(min(1 - 1 / (it + 100), 0.999), abs=1e-12)
    s = _bare_ssl(ema=dict(ema_decay=0.99))
    s.iter = 3
    assert s._get_curr_ema_decay() == 0.99
    # rampup_start_decay 0.5 -> start iter max(round(2), 2) = 2
    s = _bare_ssl(ema=dict(ema_decay=0.999, true_avg_rampup=True, rampup_start_decay=0.5))
    s.iter = 0
    assert s._get_curr_ema_decay() == pytest.approx(0.5)


def test_ssl_weight_ramp():
    """ssl.py:165-181: w * exp(-5 (1 - t)^2), 0 before the start iteration."""
    s = _bare_ssl(weight=dict(weight=2.0, weight_rampup_start_iter=10, weight_rampup_num_iter=100))
    s.iter = 5
    assert s._get_curr_ssl_weight() == 0.0
    s.iter = 10
    assert s._get_curr_ssl_weight() == pytest.approx(2.0 * math.exp(-5.0))
    s.iter = 60
    assert s._get_curr_ssl_weight() == pytest.approx(2.0 * math.exp(-5.0 * 0.25))
    s.iter = 500
    assert s._get_curr_ssl_weight() == pytest.approx(2.0)
    s = _bare_ssl()
    assert s._get_curr_ssl_weight() == 1


def test_loss_bookkeeping():
    s = _bare_ssl()
    a = dict(x=torch.tensor([1.0, 3.0]), y=[torch.tensor([2.0, 4.0]), torch.tensor(1.0)])
    c = s._collapse_losses(dict(a))
    assert float(c['x']) == 2.0 and float(c['y']) == 4.0
    out = s._sum_update_losses(dict(x=torch.tensor(1.0)), dict(x=torch.tensor([2.0, 4.0]), z=torch.tensor(5.0)))
    assert float(out['x']) == 4.0 and float(out['z']) == 5.0
    with pytest.raises(TypeError):
        s._collapse_losses(dict(bad=3.0))
    loss, log = s._parse_losses({'sup.stu.loss_cls': torch.tensor([1.0, 2.0]),
                                 'ssl.unlab.metrics.tea': torch.tensor(7.0),
                                 'ssl.lab.hp.loss': [torch.tensor(2.0)], 'ssl.weight': torch.tensor(1.0)})
    assert float(loss) == pytest.approx(3.5)
    assert float(log['loss']) == pytest.approx(3.5) and float(log['ssl.unlab.metrics.tea']) == 7.0


# ---------------------------------------------------------------- tensor-glue SSL modules
def test_max_score_filter_detach_average_numpreds():
    boxes = torch.arange(20.0).view(5, 4).requires_grad_()
    scores = torch.tensor([[.9, .0, .0, .1], [.05, .02, .0, .95], [.0, .3, .0, .7], [.0, .0, .0, 1.],
                           [.0, .0, .11, .0]])
    bd = dict(tea={'in': [(boxes, scores), (boxes[:0], scores[:0])]}, ssl_losses=dict())
    M.MaxScoreFilter(True, 0.1, 'tea.in', 'tea.out').forward(None, bd)
    assert bd['tea']['out'][0][0].shape[0] == 3 and bd['tea']['out'][1][0].shape[0] == 0
    with pytest.raises(Exception):   # mlvl_set refuses to overwrite
        M.MaxScoreFilter(True, 0.1, 'tea.in', 'tea.out').forward(None, bd)
    M.DetachBboxes('tea.out', 'tea.dt').forward(None, bd)
    assert not bd['tea']['dt'][0][0].requires_grad and bd['tea']['out'][0][0].requires_grad
    M.AverageBboxes_2D(True, False, 'tea.in', 'tea.in2', 'tea.avg').forward(
        None, dict(tea=bd['tea'], **{}) if bd['tea'].update({'in2': [(boxes + 2, scores[:, :3]),
                                                                  (boxes[:0], scores[:0, :3])]}) is None else None)
    assert torch.allclose(bd['tea']['avg'][0][0], boxes + 1)
    assert torch.allclose(bd['tea']['avg'][0][1], scores[:, :3])
    M.NumPreds('tea.out', 'n').forward(None, bd)
    assert float(bd['ssl_losses']['metrics.n']) == 1.5


def test_fusion_hungarian_matching_2d_to_2d():
    """project_3d_to_2d=False path: plain 2D<->2D matching, cost_thr drops bad pairs, outputs
    index-aligned."""
    mod = M.FusionHungarianMatching(
        assigner_cfg=ASSIGNER_CFG, cost_thr=-1.5, img_metas='stu.img_metas',
        cls_includes_bg_pred_3d=False, cls_includes_bg_pred_2d=True,
        in_bboxes_3d_key='stu.a', in_bboxes_2d_key='tea.b', out_bboxes_3d_key='stu.a_m',
        out_bboxes_2d_key='tea.b_m', match_cost_key='stu.cost', project_3d_to_2d=False)
    a = torch.tensor([[100., 100., 200., 200.], [600., 50., 700., 300.], [900., 10., 950., 60.]])
    sa = torch.tensor([[.9, .05, .05], [.1, .8, .1], [.3, .3, .4]])
    b = torch.tensor([[598., 52., 702., 297.], [101., 99., 199., 202.]])
    sb = torch.tensor([[.1, .9, .05, .0], [.95, .02, .02, .0]])
    meta = dict(ori_shape=(375, 1242, 3))
    bd = dict(stu=dict(a=[(a, sa), (a[:0], sa[:0])], img_metas=[meta, meta]),
              tea=dict(b=[(b, sb), (b, sb)]))
    mod.forward(None, bd)
    m3, m2 = bd['stu']['a_m'][0], bd['tea']['b_m'][0]
    assert torch.equal(m3[0], a[:2]) and torch.equal(m2[0], b[[1, 0]])
    assert torch.equal(m2[1], sb[[1, 0]]) and (bd['stu']['cost'][0] < -1.5).all()
    assert bd['stu']['a_m'][1][0].shape[0] == 0 and bd['tea']['b_m'][1][0].shape[0] == 0


def test_hungarian_consistency_losses():
    mod = M.HungarianConsistency(
        loss_cls_cfg=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25),
        loss_iou_cfg=dict(type='GIoULoss'), loss_l1_cfg=dict(type='L1Loss'),
        loss_weights_cfg=dict(cls_loss=2, l1_loss=20, iou_loss=2),
        in_bboxes_key='stu.p', target_bboxes_key='tea.t', cls_includes_bg_pred_in=False,
        cls_includes_bg_pred_target=True, target_img_metas_key='stu.img_metas', name='cons')
    p = torch.tensor([[100., 100., 200., 200.]], requires_grad=True)
    ps = torch.tensor([[.8, .1, .1]], requires_grad=True)
    t = torch.tensor([[110., 100., 200., 220.]])
    ts = torch.tensor([[.9, .05, .05, .0]])
    meta = dict(img_shape=(375, 1242, 3))
    bd = dict(stu=dict(p=[(p, ps), (p[:0], ps[:0])], img_metas=[meta, meta]),
              tea=dict(t=[(t, ts), (t[:0], ts[:0])]), ssl_losses=dict())
    bd = mod.forward(_bare_ssl(), bd)
    out = bd['ssl_losses']
    l1 = (10 / 1242 + 20 / 375) / 4 * 20
    assert float(out['cons.l1_loss'].detach()) == pytest.approx(l1, rel=1e-5)
    inter, union, enc = 90 * 100, 100 * 100 + 90 * 120 - 90 * 100, 100 * 120
    giou = inter / union - (enc - union) / enc
    assert float(out['cons.iou_loss']) == pytest.approx(2 * (1 - giou), rel=1e-5)
    assert float(out['cons.cls_loss']) > 0
    (out['cons.cls_loss'] + out['cons.l1_loss'] + out['cons.iou_loss']).backward()
    assert p.grad.abs().sum() > 0 and ps.grad.abs().sum() > 0
    # no pairs at all -> zero-valued losses that still carry requires_grad
    bd2 = dict(stu=dict(p=[(p[:0], ps[:0])], img_metas=[meta]), tea=dict(t=[(t[:0], ts[:0])]),
               ssl_losses=dict())
    out2 = mod.forward(_bare_ssl(), bd2)['ssl_losses']
    assert float(out2['cons.l1_loss']) == 0.0 and out2['cons.l1_loss'].requires_grad


# ---------------------------------------------------------------- optimizer / schedule / runner
class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.student = nn.ModuleDict(dict(detector_3d=nn.Linear(2, 2), detector_2d=nn.Linear(2, 1)))
        self.teacher = nn.ModuleDict(dict(detector_3d=nn.Linear(2, 2), detector_2d=nn.Linear(2, 1)))
        self.iter, self.epoch, self.seen = None, None, []

    def train_step(self, data, optimizer=None):
        self.seen.append(self.iter)
        x = data['lab_stu']
        loss = self.student['detector_3d'](x).square().mean() + self.student['detector_2d'](x).square().mean()
        return dict(loss=loss, log_vars=dict(loss=loss.detach()), num_samples=len(data['img_metas']))


OPT_CFG = {'constructor': 'HybridOptimizerConstructor',
           'student.detector_3d': dict(type='AdamW', lr=0.02, betas=(0.95, 0.99), weight_decay=0.01,
                                       step_interval=1),
           'student.detector_2d': dict(type='SGD', lr=0.04, momentum=0.9, weight_decay=0.0001,
                                       step_interval=2),
           'teacher': dict(type='SGD', lr=1e-9, momentum=0.9, weight_decay=0.0001, step_interval=1)}


def test_hybrid_optimizer_and_runner():
    torch.manual_seed(0)
    model = _Toy()
    opt = R.build_optimizer(model, OPT_CFG)
    assert isinstance(opt, R.HybridOptimizer) and len(opt.optimizers) == 3
    assert isinstance(opt.optimizers[0], torch.optim.AdamW) and opt.step_intervals == [1, 2, 1]
    assert sum(len(g['params']) for g in opt.optimizers[0].param_groups) == 2
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    run = R.IterBasedSSLRunner(model, optimizer=opt, max_iters=6)
    run.register_training_hooks(
        lr_config=dict(policy='step', warmup='linear', warmup_iters=4, warmup_ratio=0.001, step=[]),
        optimizer_config=dict(grad_clip=dict(max_norm=10, norm_type=2)),
        custom_hooks=[dict(type='ModelIterEpochHook'), dict(type='WandbVisHook')])
    lrs = []

    class Spy(R.Hook):
        def after_train_iter(self, runner):
            lrs.append(runner.optimizer.optimizers[0].param_groups[0]['lr'])
    run.register_hook(Spy())
    lab = [dict(stu=torch.randn(4, 2), img_metas=[0, 1])] * 2
    unlab = [dict(stu=torch.randn(4, 2), img_metas=[0, 1])] * 3
    w2d = model.student['detector_2d'].weight.detach().clone()
    run.run([lab, unlab], [('train', 2)])
    assert run.iter == 6
    # model.iter lags by one: after_train_iter runs BEFORE the runner increments its counter
    # (iter_based_ssl_runner.py:37-39), so iterations 0 and 1 both see 0
    assert model.seen == [0, 0, 1, 2, 3, 4] and model.iter == 5
    expect = [0.02 * (1 - (1 - it / 4) * (1 - 0.001)) if it < 4 else 0.02 for it in range(6)]
    assert lrs == pytest.approx(expect)
    assert opt.num_step_updated == 6 and not torch.equal(w2d, model.student['detector_2d'].weight)
    assert 'grad_norm' in run.log_buffer and len(run.log_buffer['loss']) == 6
    with pytest.raises(AssertionError):   # a parameter no optimizer key matches
        R.HybridOptimizerConstructor({'student': dict(type='SGD', lr=0.1)})(model)


def test_checkpoint_hook_and_resume(tmp_path):
    """ssl_train.py:113 (checkpoint_config -> CheckpointHook) and :157-166 (resume_from / load_from /
    load_from_with_optimizer): an interrupted run that resumes from iter_4.pth ends where the uninterrupted
    one does — weights, optimizer state and the iteration counter come back."""
    def make(max_iters, work_dir):
        torch.manual_seed(0)
        model = _Toy()
        opt = R.build_optimizer(model, OPT_CFG)
        run = R.IterBasedSSLRunner(model, optimizer=opt, max_iters=max_iters, work_dir=str(work_dir))
        run.register_training_hooks(
            lr_config=dict(policy='step', warmup='linear', warmup_iters=4, warmup_ratio=0.001, step=[]),
            optimizer_config=dict(grad_clip=dict(max_norm=10, norm_type=2)),
            checkpoint_config=dict(interval=2, by_epoch=False, max_keep_ckpts=2),
            log_config=dict(interval=50, hooks=[dict(type='TextLoggerHook')]),
            custom_hooks=[dict(type='ModelIterEpochHook')])
        return model, opt, run
    g = torch.Generator().manual_seed(1)
    lab = [dict(stu=torch.randn(4, 2, generator=g), img_metas=[0, 1]) for _ in range(2)]
    unlab = [dict(stu=torch.randn(4, 2, generator=g), img_metas=[0, 1]) for _ in range(3)]
    full_model, _, full = make(8, tmp_path / 'full')
    full.run([lab, unlab], [('train', 1)])
    assert sorted(os.listdir(tmp_path / 'full')) == ['iter_6.pth', 'iter_8.pth', 'latest.pth']    # max_keep_ckpts
    ckpt = torch.load(tmp_path / 'full' / 'latest.pth', weights_only=False)
    assert ckpt['meta']['iter'] == 8 and set(ckpt) == {'meta', 'state_dict', 'optimizer'}
    assert ckpt['optimizer']['num_step_updated'] == 8
    # interrupted after 4 iterations, resumed in a fresh process state
    _, _, first = make(4, tmp_path / 'part')
    first.run([lab, unlab], [('train', 1)])
    model, opt, second = make(8, tmp_path / 'part')
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)                           # whatever the fresh model holds is overwritten
    second.resume(str(tmp_path / 'part' / 'iter_4.pth'))
    assert second.iter == 4 and opt.num_step_updated == 4
    second.run([lab[0:2], unlab[1:] + unlab[:1]], [('train', 1)])      # the loaders restart; same batches as 4..7
    assert second.iter == 8
    # iterations 4..7 of the full run saw lab[0], lab[1], lab[0], lab[1] and unlab[1], unlab[2], unlab[0], unlab[1]
    for (n, a), (_, b) in zip(full_model.state_dict().items(), model.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), n
    # load_from: weights only, counters untouched
    m3, o3, third = make(8, tmp_path / 'third')
    third.load_checkpoint(str(tmp_path / 'full' / 'iter_8.pth'))
    assert third.iter == 0 and o3.num_step_updated == 0
    assert torch.equal(m3.student['detector_3d'].weight, full_model.student['detector_3d'].weight)


def test_checkpoint_key_layout_and_pretrained_loading():
    """the released checkpoints' key layout
    (`teacher.detector_3d.model.backbone_3d.conv1.0.0.weight`, sparse weights (kz,ky,kx,Cin,Cout),
    mmdet names for the 2D branch) and SSL._load_from_state_dict (ssl.py:102-127): a checkpoint
    without `teacher` keys is a pre-trained MMDetector and initialises BOTH teacher and student;
    one with them is loaded as is."""
    import copy
    from detmatch_amd import configs
    from detmatch_amd.mm3d import register_all
    from detmatch_amd.mm3d.registry import build_detector
    register_all()
    model = build_detector(configs.detmatch_kitti_model())
    sd = model.state_dict()
    for who in ('teacher', 'student'):
        assert tuple(sd[who + '.detector_3d.model.backbone_3d.conv1.0.0.weight'].shape) == (3, 3, 3, 16, 16)
        assert tuple(sd[who + '.detector_3d.model.backbone_3d.conv_out.0.weight'].shape) == (3, 1, 1, 64, 128)
        assert tuple(sd[who + '.detector_3d.model.roi_head.shared_fc_layer.0.weight'].shape) == (256, 27648, 1)
        for k, shape in (('backbone.conv1.weight', (64, 3, 7, 7)),
                         ('backbone.layer1.0.downsample.0.weight', (256, 64, 1, 1)),
                         ('backbone.layer4.2.conv3.weight', (2048, 512, 1, 1)),
                         ('neck.lateral_convs.0.conv.weight', (256, 256, 1, 1)),
                         ('neck.fpn_convs.3.conv.weight', (256, 256, 3, 3)),
                         ('rpn_head.rpn_conv.weight', (256, 256, 3, 3)),
                         ('rpn_head.rpn_cls.weight', (3, 256, 1, 1)),
                         ('rpn_head.rpn_reg.weight', (12, 256, 1, 1)),
                         ('roi_head.bbox_head.shared_fcs.0.weight', (1024, 12544)),
                         ('roi_head.bbox_head.fc_cls.weight', (4, 1024)),       # mmdet 2.14: num_classes + 1 logits
                         ('roi_head.bbox_head.fc_reg.weight', (12, 1024))):
            assert tuple(sd['%s.detector_2d.%s' % (who, k)].shape) == shape, k
    # pre-trained (teacher-less) checkpoint -> both copies
    pre = {k[len('student.'):]: torch.full_like(v, 0.25) if v.is_floating_point() else v.clone()
           for k, v in sd.items() if k.startswith('student.')}
    model.load_state_dict(copy.deepcopy(pre))
    after = model.state_dict()
    k = 'detector_3d.model.dense_head.conv_cls.weight'
    assert float(after['teacher.' + k].mean()) == 0.25 and float(after['student.' + k].mean()) == 0.25
    # full SSL checkpoint -> as is
    full = {k: (torch.full_like(v, 0.5 if k.startswith('teacher.') else 0.75) if v.is_floating_point() else v)
            for k, v in after.items()}
    model.load_state_dict(full)
    after = model.state_dict()
    assert float(after['teacher.' + k].mean()) == 0.5 and float(after['student.' + k].mean()) == 0.75


def test_epoch_based_runner_cyclic_and_step_schedules():
    """The pre-training recipes' driver (configs/detmatch/001/pretrain_pvrcnn, pretrain_frcnn):
    EpochBasedRunner + cyclic LR / momentum (mmcv formulas restated; closed-form check) and the
    by-epoch step policy with per-iteration linear warm-up."""
    from detmatch_amd import configs
    from detmatch_amd.mm3d.base_detector import DetectorStepMixin

    class Det(DetectorStepMixin, nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = nn.Linear(3, 2)

        def forward_train(self, x, img_metas):
            return dict(loss_a=self.fc(x).square().mean(), acc=torch.tensor(1.0), loss_list=[self.fc(x).abs().sum()])

    torch.manual_seed(0)
    loader = [dict(x=torch.randn(4, 3), img_metas=[0, 1, 2, 3]) for _ in range(5)]
    # ---- pretrain_pvrcnn: AdamW, cyclic lr (x10 up over 40 %, down to x1e-4) and momentum
    sched = configs.pretrain_pvrcnn_schedule(batch_size=8, max_epochs=4)
    model = Det()
    opt = R.build_optimizer(model, sched['optimizer'])
    run = R.build_from_cfg(dict(sched['runner'], model=model, optimizer=opt), R.RUNNERS)
    assert isinstance(run, R.EpochBasedRunner)
    run.register_training_hooks(sched['lr_config'], sched['optimizer_config'],
                                momentum_config=sched['momentum_config'])
    seen = []

    class Spy(R.Hook):
        def after_train_iter(self, runner):
            g = runner.optimizer.param_groups[0]
            seen.append((g['lr'], g['betas'][0] if 'betas' in g else g['momentum']))
    run.register_hook(Spy())
    w0 = model.fc.weight.detach().clone()
    run.run([loader], [('train', 1)])
    assert run.epoch == 4 and run.iter == 20 and run.max_iters == 20
    assert not torch.equal(w0, model.fc.weight) and len(run.log_buffer['loss']) == 20
    base, up = 0.004, 8
    cos = lambda a, b, f: b + 0.5 * (a - b) * (math.cos(math.pi * f) + 1)
    for it, (lr, mom) in enumerate(seen):
        if it < up:
            assert lr == pytest.approx(cos(base, base * 10, it / up))
            assert mom == pytest.approx(cos(0.9, 0.9 * 0.85 / 0.95, it / up))
        else:
            assert lr == pytest.approx(cos(base * 10, base * 1e-4, (it - up) / (20 - up)))
            assert mom == pytest.approx(cos(0.9 * 0.85 / 0.95, 0.9, (it - up) / (20 - up)))
    assert seen[0][0] == pytest.approx(base) and max(s[0] for s in seen) == pytest.approx(10 * base)
    # ---- pretrain_frcnn: SGD, step [8, 10] counted in EPOCHS, linear warm-up counted in iterations
    sched = configs.pretrain_frcnn_schedule(batch_size=8, max_epochs=12)
    sched['lr_config'].update(warmup_iters=7)
    model = Det()
    opt = R.build_optimizer(model, sched['optimizer'])
    run = R.EpochBasedRunner(model, optimizer=opt, max_epochs=12)
    run.register_training_hooks(sched['lr_config'], sched['optimizer_config'])
    seen.clear()
    run.register_hook(Spy())
    run.run([loader], [('train', 1)])
    assert run.iter == 60 and 'grad_norm' not in run.log_buffer            # grad_clip=None
    for it, (lr, _) in enumerate(seen):
        regular = 0.08 * 0.1 ** sum(it // 5 >= s for s in (8, 10))
        warm = 1 - (1 - it / 7) * (1 - 0.001) if it < 7 else 1.0
        assert lr == pytest.approx(regular * warm), it
    # stand-alone detector interface: loss = sum of the 'loss' keys, lists summed, metrics excluded
    out = model.train_step(loader[0])
    assert out['num_samples'] == 4 and set(out['log_vars']) == {'loss_a', 'acc', 'loss_list', 'loss'}
    assert float(out['log_vars']['loss']) == pytest.approx(float(out['log_vars']['loss_a'] + out['log_vars']['loss_list']))


def test_training_mode_shortcut_is_only_taken_in_the_steady_state():
    """The runners skip the per-iteration model.train() tree walk only when SSL.training_mode_ok():
    student training, teacher eval — exactly the state model.train() + train_step's teacher.eval() of
    the reference leave behind (iter_based_ssl_runner.py:16, ssl.py:219)."""
    from detmatch_amd import configs
    from detmatch_amd.mm3d import register_all
    from detmatch_amd.mm3d.registry import build_detector
    register_all()
    model = build_detector(configs.detmatch_kitti_model())
    calls = []
    orig = model.train
    model.train = lambda mode=True: (calls.append(mode), orig(mode))[1]
    R._ensure_train_mode(model)                    # fresh model: teacher still in training mode
    assert calls == [True] and not model.training_mode_ok()
    model.teacher.eval()                           # what train_step does
    assert model.training_mode_ok()
    R._ensure_train_mode(model)
    assert calls == [True]                         # steady state: no second walk
    model.student.detector_2d.eval()               # any deviation -> the plain call again
    assert not model.training_mode_ok()
    R._ensure_train_mode(model)
    assert calls == [True, True] and model.student.detector_2d.training
    model.eval()
    R._ensure_train_mode(model)
    assert calls == [True, True, False, True] and model.training          # eval() itself is train(False)


def test_rng_windows_make_draws_independent_of_module_order():
    """ssl._RngWindows on a generator with Philox offsets: module i always draws from offset
    base + (i + 1) * 2^32, whatever was drawn before, and the iteration ends at base + (n + 1) * 2^32."""
    import torch
    from detmatch_amd.mm3d import ssl

    class Gen(object):                     # stands in for the device generator (CPU has no offsets)
        def __init__(self):
            self.off = 40

        def get_offset(self):
            return self.off

        def set_offset(self, v):
            self.off = v

    mods = [object(), object(), object()]
    w = ssl._RngWindows(torch.device('cpu'), mods)
    assert w.gen is None and w.offset() is None        # CPU: a no-op
    w.gen, w.base = Gen(), 40
    W = ssl._RngWindows.WINDOW
    w.enter(mods[2])
    assert w.gen.off == 40 + 3 * W
    w.gen.off += 12                                    # module 2 drew something
    saved = w.offset()
    w.enter(mods[0])                                   # a module issued from inside another one
    assert w.gen.off == 40 + 1 * W
    w.restore(saved)
    assert w.gen.off == 40 + 3 * W + 12
    w.enter(object())                                  # unknown module: untouched
    assert w.gen.off == 40 + 3 * W + 12
    w.finish()
    assert w.gen.off == 40 + 4 * W


def test_lazy_log_vars_compute_on_first_read_only():
    """base_detector._parse_losses: the loss comes back at once, the logged values are packed when somebody reads them
    (the runner reads them after it has issued the backward pass)."""
    from detmatch_amd.mm3d.base_detector import DetectorStepMixin, LazyLogVars

    class _M(DetectorStepMixin):
        pass
    a = torch.tensor([1.0, 3.0], requires_grad=True)
    losses = {'loss_a': a * 2, 'loss_b': [a.sum(), a.sum() * 0.5], 'acc': torch.tensor(0.25)}
    loss, logs = _M()._parse_losses(losses)
    assert isinstance(logs, LazyLogVars) and logs._pending is not None            # nothing packed yet
    assert float(loss) == pytest.approx(4.0 + 6.0) and loss.requires_grad          # mean(2, 6) + (4 + 2)
    logs.update({'extra': torch.tensor(7.0)})                                      # merged without forcing the values
    assert logs._pending is not None
    loss.backward()
    assert torch.allclose(a.grad, torch.tensor([2.5, 2.5]))
    assert float(logs['loss']) == pytest.approx(10.0) and logs._pending is None    # first read packs everything
    assert float(logs['loss_a']) == pytest.approx(4.0) and float(logs['acc']) == 0.25 and float(logs['extra']) == 7.0
    assert set(logs.keys()) == {'loss_a', 'loss_b', 'acc', 'loss', 'extra'} and len(logs) == 5
    assert not logs['loss'].requires_grad


def test_runner_buffers_lazy_logs_when_somebody_looks():
    from detmatch_amd.mm3d.base_detector import LazyLogVars
    from detmatch_amd.mm3d import runner as R
    run = R.IterBasedSSLRunner.__new__(R.IterBasedSSLRunner)
    run._lazy_logs, run._log_buffer = [], {}
    for k in range(3):
        lv = LazyLogVars({'loss_x': torch.tensor(float(k))}, torch.tensor(float(k)))
        run._after_step(dict(loss=torch.tensor(0.0), log_vars=lv))
        assert lv._pending is not None and len(run._lazy_logs) == k + 1           # not computed, not buffered yet
    buf = run.log_buffer
    assert [float(v) for v in buf['loss_x']] == [0.0, 1.0, 2.0] and not run._lazy_logs
    run._after_step(dict(loss=torch.tensor(0.0), log_vars={'plain': torch.tensor(1.0)}))
    assert 'plain' in run.log_buffer


def test_hybrid_optimizer_skips_members_that_cannot_have_gradients():
    """The recipe's 'teacher' SGD holds frozen parameters only: torch's step() would walk 378 one-parameter groups to
    find nothing to do."""
    from detmatch_amd.mm3d.runner import HybridOptimizer
    w = torch.nn.Parameter(torch.ones(3))
    frozen = [torch.nn.Parameter(torch.ones(2), requires_grad=False) for _ in range(4)]
    o1 = torch.optim.SGD([w], lr=0.5)
    o2 = torch.optim.SGD([dict(params=[p]) for p in frozen], lr=0.5)
    calls = []
    o2.step = lambda *a, **k: calls.append(1)
    opt = HybridOptimizer.__new__(HybridOptimizer)
    opt.optimizers, opt.step_intervals, opt.num_step_updated = [o1, o2], [1, 1], 0
    w.grad = torch.ones(3)
    opt.step()
    assert torch.allclose(w.detach(), torch.full((3,), 0.5)) and not calls
    frozen[0].grad = torch.ones(2)            # somebody handed it a gradient after all: torch decides
    opt.step()
    assert calls == [1]


def test_runner_draws_one_batch_ahead_in_the_same_order_and_not_beyond_the_end():
    """IterBasedSSLRunner.draw_ahead (what lets SSL._forward_train start the geometry / the teacher's 2D pass under the
    previous iteration's tail): batch i + 1 is drawn before iteration i is issued — the model sees the same batches in
    the same order as without, and nothing is drawn for an iteration that will not run."""
    class Counting(list):
        draws = 0

        def __iter__(self):
            for item in list.__iter__(self):
                type(self).draws += 1
                yield item

    class Lab(Counting):
        pass

    class Unlab(Counting):
        pass

    seen = {}
    for ahead in (False, True):
        torch.manual_seed(0)
        model = _Toy()
        model.prefetch_geometry = lambda *a, **k: None
        got = []
        step = model.train_step
        model.train_step = lambda data, optimizer=None, got=got, step=step: (got.append(float(data['lab_stu'][0, 0])),
                                                                              step(data, optimizer))[1]
        run = R.IterBasedSSLRunner(model, optimizer=R.build_optimizer(model, OPT_CFG), max_iters=5)
        run.lookahead, run.draw_ahead = False, ahead
        run.register_training_hooks(lr_config=dict(policy='step', step=[]), optimizer_config=dict(grad_clip=None))
        epochs = []

        class Spy(R.Hook):
            def after_train_iter(self, runner):
                epochs.append(runner.epoch)
        run.register_hook(Spy())
        Lab.draws = Unlab.draws = 0
        lab = Lab(dict(stu=torch.full((4, 2), float(i)), img_metas=[0, 1]) for i in range(3))
        unlab = Unlab(dict(stu=torch.full((4, 2), float(10 + i)), img_metas=[0, 1]) for i in range(4))
        run.run([lab, unlab], [('train', 1)])
        assert run.iter == 5
        seen[ahead] = (got, Lab.draws, Unlab.draws, epochs)
        assert not hasattr(model, '_data_ready') or not torch.cuda.is_available()
    assert seen[True][0] == seen[False][0] == [0.0, 1.0, 2.0, 0.0, 1.0]
    assert seen[True][1:3] == seen[False][1:3] == (5, 5)
    # runner.epoch is the labeled loader's epoch BEFORE the iteration's batch was drawn (IterBasedRunner.train), ahead or not
    assert seen[True][3] == seen[False][3] == [0, 0, 0, 0, 1]


def test_issue_order_is_decided_in_one_place(monkeypatch):
    """mm3d/schedule.py: the config-driven entry and the bench workload call the same function; it leaves non-SSL models and
    plain (non-FlatGradDDP) wrappers alone, and switches the look-ahead draw on only together with the lanes."""
    from detmatch_amd.mm3d.parallel import FlatGradDDP
    from detmatch_amd.mm3d.schedule import apply_issue_order

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Linear(2, 2)

        def _forward_train(self):
            pass

    plain = nn.Linear(2, 2)
    assert apply_issue_order(plain, FlatGradDDP(plain, broadcast=False)) is False and not hasattr(plain, 'two_lanes')
    m = Fake()
    assert apply_issue_order(m, m) is False                     # not wrapped: one stream, one backward pass
    for lanes in ('1', '0'):
        monkeypatch.setenv('DM_TWO_LANES', lanes)
        monkeypatch.delenv('DM_LOOKAHEAD', raising=False)
        m = Fake()
        ddp = FlatGradDDP(m, broadcast=False)
        run = R.IterBasedSSLRunner(ddp, optimizer=torch.optim.SGD(m.parameters(), lr=0.1), max_iters=1)
        before = run.lookahead
        assert apply_issue_order(m, ddp) and apply_issue_order(m, ddp, run)
        assert m.early_backward and m.side_wgrad and m.share_2d_trunk and m.after_partial_backward == ddp.collect
        assert m.two_lanes == (lanes == '1') and m.lane_mode == (None if lanes == '1' else 'glue')
        assert run.draw_ahead == (lanes == '1') and run.lookahead == (False if lanes == '1' else before)


def test_text_logger_writes_one_line_per_interval(capsys):
    """runner.TextLoggerHook from log_config (split_0.py: log_config = dict(interval=50, hooks=[TextLoggerHook, ...])): window
    means of the logged values, the learning rates, behind every other hook."""
    torch.manual_seed(0)
    model = _Toy()
    run = R.IterBasedSSLRunner(model, optimizer=R.build_optimizer(model, OPT_CFG), max_iters=4)
    run.register_training_hooks(lr_config=dict(policy='step', step=[]), optimizer_config=dict(grad_clip=dict(max_norm=10, norm_type=2)),
                                log_config=dict(interval=2, hooks=[dict(type='TextLoggerHook'), dict(type='WandbLoggerHook')]),
                                custom_hooks=[dict(type='ModelIterEpochHook')])
    lab = [dict(stu=torch.randn(4, 2), img_metas=[0, 1])] * 2
    run.run([lab, lab], [('train', 1)])
    names = [type(h).__name__ for h in run._hooks]
    assert names[-2:] == ['TextLoggerHook', 'WandbLoggerHook'] and names.index('OptimizerHook') < names.index('TextLoggerHook')
    lines = run._hooks[-2].lines
    assert len(lines) == 2 and lines[0].startswith('Iter [2/4]') and lines[1].startswith('Iter [4/4]')
    want = float(torch.stack([v.float() for v in run.log_buffer['loss'][2:4]]).mean())
    assert 'loss: %.4f' % want in lines[1] and 'grad_norm: ' in lines[1] and 'lr: ' in lines[1] and 'time: ' in lines[1]
    assert lines[1] in capsys.readouterr().out
