"""Step driver: IterBasedSSLRunner, HybridOptimizer(+Constructor), ModelIterEpochHook, the
optimizer hook (grad clip) and the warm-up LR schedule — mmdet3d/core/runner/
iter_based_ssl_runner.py:11-110, core/optimizer/hybrid_{constructor,optimizer}.py,
core/utils/model_iter_epoch.py:18-28, configs/detmatch/001/detmatch/split_0.py:829-862.

mmcv's IterBasedRunner / hook machinery (un-vendored, mmcv-full 1.3.16) is restated only as far
as the training iteration needs it: parity unpinned for the mmcv parts.
"""
import copy
import math
import os
import time

import torch
import torch.nn as nn

from .parallel import FlatGradDDP
from .registry import HOOKS, OPTIMIZER_BUILDERS, OPTIMIZERS, RUNNERS, build_from_cfg

OPTIMIZERS.register_module(torch.optim.AdamW, name='AdamW')
OPTIMIZERS.register_module(torch.optim.SGD, name='SGD')
OPTIMIZERS.register_module(torch.optim.Adam, name='Adam')


@OPTIMIZERS.register_module()
class HybridOptimizer(torch.optim.Optimizer):
    """hybrid_optimizer.py:5-101: several optimizers, each stepped every step_interval calls."""

    def __init__(self, optimizers, step_intervals=None):
        self.optimizers = optimizers
        self.param_groups = []
        for o in optimizers:
            self.param_groups += o.param_groups
        if not isinstance(step_intervals, list):
            step_intervals = [1] * len(optimizers)
        self.step_intervals = step_intervals
        self.num_step_updated = 0

    def __repr__(self):
        s = self.__class__.__name__ + ' (\n'
        for o, k in zip(self.optimizers, self.step_intervals):
            s += 'Update interval: %d\n' % k + repr(o).replace('\n', '\n  ') + ',\n'
        return s + ')'

    def state_dict(self):
        for f in getattr(self, '_fused', None) or []:
            if f is not None:     # fresh per-parameter step tensors, no state for never-used params
                f.sync_published_state()
        sds = [o.state_dict() for o in self.optimizers]
        return dict(num_step_updated=self.num_step_updated, state=[s['state'] for s in sds],
                    param_groups=[s['param_groups'] for s in sds])

    def load_state_dict(self, state_dict):
        assert len(state_dict['state']) == len(self.optimizers)
        assert len(state_dict['param_groups']) == len(self.optimizers)
        for o, st, pg in zip(self.optimizers, state_dict['state'], state_dict['param_groups']):
            o.load_state_dict(dict(state=st, param_groups=pg))
        self.param_groups = []
        for o in self.optimizers:
            self.param_groups += o.param_groups
        self.num_step_updated = state_dict['num_step_updated']
        for f in getattr(self, '_fused', None) or []:
            if f is not None:     # torch re-homed the loaded state: move it back into the flat state
                f.adopt_loaded_state()

    def zero_grad(self, set_to_none=False):
        for o in self.optimizers:
            o.zero_grad(set_to_none=set_to_none)

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.num_step_updated += 1
        fused = getattr(self, '_fused', None)
        for i, (k, o) in enumerate(zip(self.step_intervals, self.optimizers)):
            if self.num_step_updated % k == 0:
                if fused is not None and fused[i] is not None:
                    fused[i].step(getattr(self, 'grad_scale', None))
                elif self._inert(i, o):
                    continue         # no parameter of this member can ever hold a gradient (the recipe's 'teacher' SGD:
                    #                  378 one-parameter groups that torch would walk for 1.4 ms to find nothing to do)
                else:
                    ddp = getattr(self, '_ddp', None)
                    hidden = []
                    if ddp is not None:      # never-used parameters: .grad None, as in the reference
                        dead = set(id(p) for p in ddp.dead_params())
                        for g in o.param_groups:
                            for p in g['params']:
                                if id(p) in dead and p.grad is not None:
                                    hidden.append((p, p.grad))
                                    p.grad = None
                    o.step()
                    for p, g in hidden:      # the arena views stay in place (hooks mode accumulates into them)
                        p.grad = g
        self.grad_scale = None
        return loss

    def _inert(self, i, o):
        """True when no parameter of member i requires a gradient or holds one (torch's step() would skip them all)."""
        cache = self.__dict__.setdefault('_inert_cache', {})
        key = (i, sum(len(g['params']) for g in o.param_groups))
        if cache.get('key%d' % i) != key:
            cache['key%d' % i] = key
            cache[i] = [p for g in o.param_groups for p in g['params']]
        ps = cache[i]
        return not any(p.requires_grad or p.grad is not None for p in ps)

    def enable_fused(self, ddp):
        """Route AdamW / SGD members whose parameters form one contiguous range of `ddp`'s flat
        arenas through the fused kernels (dm_adamw_step_f32 / dm_sgd_step_f32).  Optimizer state
        stays visible through `optimizer.state[p]` as views of the flat state.  Returns the number
        of fused members."""
        self._ddp = ddp
        self._fused = [FusedRange.try_build(o, ddp) for o in self.optimizers]
        return sum(f is not None for f in self._fused)

    def add_param_group(self, param_group):
        raise NotImplementedError


class FusedRange(object):
    """One torch optimizer (AdamW or SGD, uniform hyper-parameters) over one contiguous range of a
    FlatGradDDP's parameter / gradient arenas."""

    def __init__(self, opt, ddp, lo, hi, params):
        self.opt, self.ddp, self.lo, self.hi = opt, ddp, lo, hi
        self.kind = 'adamw' if isinstance(opt, torch.optim.AdamW) else 'sgd'
        dev = ddp.flat.device
        self.t = 0
        n = hi - lo
        if self.kind == 'adamw':
            self.m = torch.zeros(n, device=dev)
            self.v = torch.zeros(n, device=dev)
        else:
            self.buf = torch.zeros(n, device=dev)
        self.step_t = torch.zeros((), dtype=torch.float32)
        self.params = list(params)
        # first[block] = optimizer step of the parameter's first gradient (0 = none yet): torch.optim counts
        # steps / creates momentum buffers PER PARAMETER.  Dropped (scalar step for everybody) once every
        # parameter is live and none of them started late.
        self.first = torch.zeros((n + 3) // 4, dtype=torch.int32, device=dev)
        if any(opt.state.get(p, None) for p in self.params):
            self.adopt_loaded_state()      # built after a resume: keep the loaded state
        else:
            self._publish()

    def _publish(self):
        """torch-compatible view of the state (state_dict / resume parity, hybrid_optimizer.py:41-68)."""
        opt, ddp, lo = self.opt, self.ddp, self.lo
        for p in self.params:
            o = ddp.offset[id(p)] - lo
            if self.kind == 'adamw':
                opt.state[p] = dict(step=self.step_t.clone(), exp_avg=self.m[o:o + p.numel()].view_as(p),
                                    exp_avg_sq=self.v[o:o + p.numel()].view_as(p))
            else:
                opt.state[p] = dict(momentum_buffer=self.buf[o:o + p.numel()].view_as(p))

    def sync_published_state(self):
        """Before state_dict(): every parameter gets its OWN 0-d step tensor holding the current
        count (a shared tensor would be stepped once per parameter by a plain torch optimizer that
        loads the checkpoint), and parameters that never received a gradient have no state at all,
        as in torch."""
        self.step_t.fill_(self.t)
        self._publish()
        for p in self.ddp.dead_params():
            self.opt.state.pop(p, None)
        if self.first is not None and self.kind == 'adamw':
            first = self.first.cpu()
            for p in self.params:
                st = self.opt.state.get(p, None)
                f = int(first[(self.ddp.offset[id(p)] - self.lo) // 4])
                if st and f > 0:
                    st['step'] = torch.tensor(float(self.t - f + 1))

    @torch.no_grad()
    def adopt_loaded_state(self):
        """After optimizer.load_state_dict (which replaces the per-parameter state tensors by
        copies of the checkpoint's): copy them into the flat state the kernels walk and publish the
        views again.  Parameters without loaded state (never stepped) keep zeros."""
        lo, steps = self.lo, []
        for p in self.params:
            st = self.opt.state.get(p, None)
            if not st:
                continue
            o = self.ddp.offset[id(p)] - lo
            if self.kind == 'adamw':
                self.m[o:o + p.numel()].copy_(st['exp_avg'].reshape(-1))
                self.v[o:o + p.numel()].copy_(st['exp_avg_sq'].reshape(-1))
                steps.append(int(st['step']))
            elif st.get('momentum_buffer', None) is not None:
                self.buf[o:o + p.numel()].copy_(st['momentum_buffer'].reshape(-1))
                steps.append(1)           # momentum buffer exists: not the first step any more
        if steps:
            # per-parameter step counts: the range's count is the largest, a parameter that started late
            # has first = t - step + 1
            self.t = max(steps)
            self.step_t.fill_(self.t)
            first = torch.zeros(self.first.numel() if self.first is not None else (self.hi - lo + 3) // 4,
                                dtype=torch.int32)
            for p in self.params:
                st = self.opt.state.get(p, None)
                if not st:
                    continue
                o = (self.ddp.offset[id(p)] - lo) // 4
                step = int(st['step']) if self.kind == 'adamw' else self.t
                first[o:o + (p.numel() + 3) // 4] = self.t - step + 1
            self.first = first.to(self.m.device if self.kind == 'adamw' else self.buf.device)
        self._publish()

    @staticmethod
    def try_build(opt, ddp):
        if ddp.flat_params is None or not ddp.flat.is_cuda or not ddp.check_param_arena():
            return None
        if not isinstance(opt, (torch.optim.AdamW, torch.optim.SGD)):
            return None
        params = [p for g in opt.param_groups for p in g['params'] if id(p) in ddp.offset]
        rest = [p for g in opt.param_groups for p in g['params'] if id(p) not in ddp.offset]
        if not params or any(p.requires_grad for p in rest):
            return None
        pad = lambda n: (n + 3) // 4 * 4
        lo = min(ddp.offset[id(p)] for p in params)
        hi = max(ddp.offset[id(p)] + pad(p.numel()) for p in params)
        if sum(pad(p.numel()) for p in params) != hi - lo:
            return None          # not one contiguous range
        keys = ('lr', 'weight_decay', 'betas', 'eps', 'momentum', 'dampening', 'nesterov', 'amsgrad',
                'maximize')
        g0 = opt.param_groups[0]
        for g in opt.param_groups:
            if any(g.get(k) != g0.get(k) for k in keys):
                return None      # per-parameter options (lr_mult ...): keep the torch path
        if g0.get('amsgrad') or g0.get('nesterov') or g0.get('maximize'):
            return None
        return FusedRange(opt, ddp, lo, hi, params)

    def step(self, grad_scale=None):
        from .. import _lib, dense_conv
        fp = self.ddp.flat_params     # raw-pointer update: packed copies of THESE weights are stale now
        dense_conv.weights_changed(fp.data_ptr() + 4 * self.lo, fp.data_ptr() + 4 * self.hi)
        L = _lib.lib()
        g0 = self.opt.param_groups[0]
        d = self.ddp
        p = d.flat_params[self.lo:self.hi]
        g = d.flat[self.lo:self.hi]
        gs = _lib.ptr(grad_scale) if grad_scale is not None else None
        self.t += 1
        self.step_t.fill_(self.t)
        n = self.hi - self.lo
        # parameters that never received a gradient are skipped, as torch.optim skips .grad None
        mask = d.live_mask(self.lo, self.hi)
        mk = _lib.ptr(mask) if mask is not None else None
        if self.first is not None:
            if mask is not None:       # blocks that turned live this step start counting now
                torch.where((self.first == 0) & (mask > 0), torch.full_like(self.first, self.t), self.first,
                            out=self.first)
            elif d._all_live:          # every parameter is live: keep the vector only if somebody started late
                self.first.masked_fill_(self.first == 0, self.t)
                if not bool((self.first > 1).any()):      # one read-back, once
                    self.first = None
        fs = _lib.ptr(self.first) if self.first is not None else None
        if self.kind == 'adamw':
            _lib.check(L.dm_adamw_step_blocks_f32(
                _lib.ptr(p), _lib.ptr(g), _lib.ptr(self.m), _lib.ptr(self.v), n, g0['lr'],
                g0['betas'][0], g0['betas'][1], g0['eps'], g0['weight_decay'], self.t, gs, mk, fs,
                _lib.stream()), 'dm_adamw_step_blocks_f32')
        else:
            _lib.check(L.dm_sgd_step_blocks_f32(
                _lib.ptr(p), _lib.ptr(g), _lib.ptr(self.buf), n, g0['lr'], g0['momentum'],
                g0['dampening'], g0['weight_decay'], self.t, gs, mk, fs, _lib.stream()),
                'dm_sgd_step_blocks_f32')


@OPTIMIZER_BUILDERS.register_module()
class HybridOptimizerConstructor(object):
    """hybrid_constructor.py:8-130: each top-level key of optimizer_cfg is a SUBSTRING of the
    parameter names it owns (first match in key order wins); every parameter must match."""

    def __init__(self, optimizer_cfg, paramwise_cfg=None):
        if not isinstance(optimizer_cfg, dict):
            raise TypeError('optimizer_cfg should be a dict', 'but got %s' % type(optimizer_cfg))
        self.paramwise_cfg = {} if paramwise_cfg is None else paramwise_cfg
        self.optimizer_cfg = optimizer_cfg
        self.base_lr = {k: optimizer_cfg[k].get('lr', None) for k in optimizer_cfg}

    def __call__(self, model):
        if hasattr(model, 'module'):
            model = model.module
        custom = self.paramwise_cfg.get('custom_keys', {})
        custom_sorted = sorted(sorted(custom.keys()), key=len, reverse=True)
        cfg = copy.deepcopy(dict(self.optimizer_cfg))
        groups = {k: [] for k in cfg}
        for name, p in model.named_parameters():
            owner = next((k for k in cfg if k in name), None)
            assert owner is not None, 'key %s is not matched to any optimizer' % name
            g = {'params': [p]}
            ck = next((c for c in custom_sorted if c in name), None)
            if ck is not None:
                g['lr'] = self.base_lr[owner] * custom[ck].get('lr_mult', 1.)
            groups[owner].append(g)
        opts, intervals = [], []
        for k, single in cfg.items():
            single = dict(single)
            intervals.append(single.pop('step_interval', 1))
            if not groups[k]:       # e.g. no 2D branch built: keep the slot, nothing to step
                groups[k] = [{'params': [nn.Parameter(torch.zeros(()), requires_grad=False)]}]
            single['params'] = groups[k]
            if single.get('type') in ('AdamW', 'Adam') and 'betas' in single:
                single['betas'] = tuple(single['betas'])
            opts.append(build_from_cfg(single, OPTIMIZERS))
        return HybridOptimizer(opts, intervals)


def build_optimizer(model, cfg):
    """mmcv build_optimizer: optional 'constructor' / 'paramwise_cfg' keys."""
    cfg = copy.deepcopy(dict(cfg))
    ctor = cfg.pop('constructor', None)
    paramwise = cfg.pop('paramwise_cfg', None)
    if ctor is None:
        cfg['params'] = [p for p in model.parameters() if p.requires_grad]
        return build_from_cfg(cfg, OPTIMIZERS)
    return build_from_cfg(dict(type=ctor, optimizer_cfg=cfg, paramwise_cfg=paramwise),
                          OPTIMIZER_BUILDERS)(model)


# ------------------------------------------------------------------ hooks
class Hook(object):
    def before_run(self, runner): pass
    def after_run(self, runner): pass
    def before_epoch(self, runner): pass
    def after_epoch(self, runner): pass
    def before_train_iter(self, runner): pass
    def after_train_iter(self, runner): pass
    def after_train_epoch(self, runner): pass


def _inner(model):
    return model.module if hasattr(model, 'module') else model


@HOOKS.register_module()
class ModelIterEpochHook(Hook):
    """model_iter_epoch.py:18-28: model.iter = runner.iter in before_run and in after_train_iter,
    which fires BEFORE the runner increments its counter — iterations 0 and 1 both see 0, then
    iteration k sees k - 1 (the EMA / ssl-weight schedules inherit this lag); epoch - 1 at start."""

    def before_run(self, runner):
        m = _inner(runner.model)
        m.iter = runner.iter
        m.epoch = runner.epoch - 1

    def after_train_iter(self, runner):
        _inner(runner.model).iter = runner.iter

    def after_train_epoch(self, runner):
        _inner(runner.model).epoch = runner.epoch


@HOOKS.register_module()
class OptimizerHook(Hook):
    """mmcv OptimizerHook(grad_clip=dict(max_norm, norm_type)): zero_grad, backward, clip, step.
    With a FlatGradDDP-wrapped model the gradient exchange is finished between backward and clip."""

    def __init__(self, grad_clip=None):
        self.grad_clip = grad_clip

    @staticmethod
    def _early(runner):
        return isinstance(runner.model, FlatGradDDP) and getattr(_inner(runner.model), 'early_backward', False)

    def before_train_iter(self, runner):
        if self._early(runner):      # the model back-propagates part of the loss inside train_step
            runner.model.zero_grad(arm=False)

    def after_train_iter(self, runner):
        ddp = runner.model if isinstance(runner.model, FlatGradDDP) else None
        if ddp is not None and self._early(runner):
            ddp.arm()
        elif ddp is not None:
            ddp.zero_grad()
        else:
            runner.optimizer.zero_grad()
        if runner.outputs['loss'].requires_grad:     # else: every term was back-propagated inside train_step
            from ..spconv.ops import deferred_weight_grads
            with deferred_weight_grads():      # the pass's sparse weight gradients in one launch pair
                runner.outputs['loss'].backward()
        deferred = getattr(_inner(runner.model), 'finish_deferred_backward', None)
        if deferred is not None:      # e.g. the shared 2D trunk of SSL: its backward runs once, after all the heads'
            deferred()
        if ddp is not None:
            ddp.finish()
        if self.grad_clip is not None:
            fused = getattr(runner.optimizer, '_fused', None)
            if ddp is not None and ddp.covers_all_clipped and fused and all(
                    f is not None or not any(p.requires_grad for g in o.param_groups for p in g['params'])
                    for f, o in zip(fused, runner.optimizer.optimizers)):
                # every trainable parameter is stepped by a fused kernel: hand the clip
                # coefficient to the kernels instead of rescaling the arena
                norm, runner.optimizer.grad_scale = ddp.clip_coef(**self.grad_clip)
            elif ddp is not None and ddp.covers_all_clipped:
                norm = ddp.clip_grad_norm_(**self.grad_clip)
            else:
                params = [p for g in runner.optimizer.param_groups for p in g['params']
                          if p.requires_grad and p.grad is not None]
                norm = torch.nn.utils.clip_grad_norm_(params, **self.grad_clip) if params else None
            if norm is not None:
                runner.log_buffer.setdefault('grad_norm', []).append(norm.detach())
        runner.optimizer.step()


@HOOKS.register_module()
class EvalHook(Hook):
    """mmdet EvalHook / DistEvalHook (mmdet/core/evaluation/eval_hooks.py) as ssl_train.py:132-140 registers them: every
    `interval` iterations (epochs with by_epoch) run the validation loader through the model in eval mode, hand the results
    to `dataset.evaluate(results, logger=..., **eval_kwargs)` on rank 0 and put the metrics into the log buffer.  The
    loader shards the frames over the ranks itself (datasets.KittiTestLoader); the runner puts the model back into
    training mode at the start of the next iteration."""

    def __init__(self, dataloader, start=None, interval=1, by_epoch=True, save_best=None, rule=None, **eval_kwargs):
        assert save_best is None, 'save_best is not part of the DetMatch recipes'
        self.dataloader, self.start, self.interval, self.by_epoch = dataloader, start, int(interval), by_epoch
        self.eval_kwargs = eval_kwargs
        self.last = None          # metrics of the latest evaluation (rank 0)

    def _due(self, n):
        if self.start is not None and n < self.start:
            return False
        return n % self.interval == 0

    def _evaluate(self, runner):
        from .datasets import multi_gpu_test
        results = multi_gpu_test(runner.model, self.dataloader)
        if results is None:           # not rank 0
            return
        res = self.dataloader.dataset.evaluate(results, logger=getattr(runner, 'logger', None), **self.eval_kwargs)
        self.last = dict(res)
        buf = runner.log_buffer
        for k, v in res.items():
            buf.setdefault(k, []).append(torch.as_tensor(float(v)))

    def after_train_iter(self, runner):
        if not self.by_epoch and self._due(runner.iter + 1):
            self._evaluate(runner)

    def after_train_epoch(self, runner):
        if self.by_epoch and self._due(runner.epoch + 1):
            self._evaluate(runner)


@HOOKS.register_module()
class StepLrUpdaterHook(Hook):
    """mmcv LrUpdaterHook, policy='step' (by_epoch False) with warmup: regular lr =
    base * gamma^(#steps passed); during the first warmup_iters iterations
    linear: lr * (1 - (1 - it/warmup_iters) (1 - warmup_ratio)); constant: lr * ratio;
    exp: lr * ratio^(1 - it/warmup_iters)."""

    def __init__(self, step=(), gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1,
                 by_epoch=False, **kwargs):
        self.step = [step] if isinstance(step, int) else list(step)
        self.gamma = gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.by_epoch = by_epoch       # `step` counts epochs (pretrain_frcnn: step=[8, 10]); warm-up stays per iteration
        self.base_lr = None

    def before_run(self, runner):
        for g in runner.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in runner.optimizer.param_groups]

    def get_lr(self, it, epoch=0):
        progress = epoch if self.by_epoch else it
        exp = sum(1 for s in self.step if progress >= s)
        regular = [lr * self.gamma ** exp for lr in self.base_lr]
        if self.warmup is None or it >= self.warmup_iters:
            return regular
        if self.warmup == 'constant':
            return [lr * self.warmup_ratio for lr in regular]
        if self.warmup == 'linear':
            k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in regular]
        if self.warmup == 'exp':
            k = self.warmup_ratio ** (1 - it / self.warmup_iters)
            return [lr * k for lr in regular]
        raise ValueError(self.warmup)

    def before_train_iter(self, runner):
        for g, lr in zip(runner.optimizer.param_groups, self.get_lr(runner.iter, runner.epoch)):
            g['lr'] = lr


def annealing_cos(start, end, factor, weight=1):
    """mmcv.runner.hooks.lr_updater.annealing_cos"""
    return end + 0.5 * weight * (start - end) * (math.cos(math.pi * factor) + 1)


class _CyclicHook(Hook):
    """mmcv Cyclic{Lr,Momentum}UpdaterHook (1.3.16, by_epoch=False, anneal 'cos'; un-vendored: parity
    unpinned): per cycle of max_iters // cyclic_times iterations the value goes base ->
    base*target_ratio[0] over the first step_ratio_up of the cycle, then -> base*target_ratio[1]."""

    def __init__(self, by_epoch=False, target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4,
                 anneal_strategy='cos', **kwargs):
        assert not by_epoch, 'currently only support "by_epoch" = False'
        if isinstance(target_ratio, (int, float)):
            target_ratio = (target_ratio, target_ratio / 1e5)
        assert len(target_ratio) == 2 and 0 <= step_ratio_up < 1.0
        assert anneal_strategy == 'cos'
        self.target_ratio, self.cyclic_times, self.step_ratio_up = tuple(target_ratio), cyclic_times, step_ratio_up
        self.phases = []
        self.base = None

    def before_run(self, runner):
        per = runner.max_iters // self.cyclic_times
        up = int(self.step_ratio_up * per)
        self.phases = [(0, up, per, 1, self.target_ratio[0]),
                       (up, per, per, self.target_ratio[0], self.target_ratio[1])]

    def value(self, it, base):
        for start, end, per, r0, r1 in self.phases:
            it %= per
            if start <= it < end:
                return annealing_cos(base * r0, base * r1, (it - start) / (end - start))
        return base


@HOOKS.register_module()
class CyclicLrUpdaterHook(_CyclicHook):

    def before_run(self, runner):
        super().before_run(runner)
        for g in runner.optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base = [g['initial_lr'] for g in runner.optimizer.param_groups]

    def before_train_iter(self, runner):
        for g, b in zip(runner.optimizer.param_groups, self.base):
            g['lr'] = self.value(runner.iter, b)


@HOOKS.register_module()
class CyclicMomentumUpdaterHook(_CyclicHook):
    """target_ratio default (0.85/0.95, 1); drives `momentum` (SGD) or betas[0] (Adam family)."""

    def __init__(self, target_ratio=(0.85 / 0.95, 1), **kwargs):
        super().__init__(target_ratio=target_ratio, **kwargs)

    def before_run(self, runner):
        super().before_run(runner)
        self.base = []
        for g in runner.optimizer.param_groups:
            if 'momentum' in g:
                g.setdefault('initial_momentum', g['momentum'])
            else:
                g.setdefault('initial_momentum', g['betas'][0])
            self.base.append(g['initial_momentum'])

    def before_train_iter(self, runner):
        for g, b in zip(runner.optimizer.param_groups, self.base):
            m = self.value(runner.iter, b)
            if 'momentum' in g:
                g['momentum'] = m
            else:
                g['betas'] = (m, g['betas'][1])


@HOOKS.register_module()
class CheckpointHook(Hook):
    """mmcv CheckpointHook (1.3.16; un-vendored: parity unpinned): every `interval` iterations
    (by_epoch=False -> iter_{n}.pth) or epochs (epoch_{n}.pth) rank 0 writes model + optimizer + meta and
    points latest.pth at it; `max_keep_ckpts` > 0 removes older files (configs/detmatch/*: interval 5000,
    by_epoch False)."""

    def __init__(self, interval=-1, by_epoch=True, save_optimizer=True, out_dir=None, max_keep_ckpts=-1,
                 save_last=True, **kwargs):
        self.interval, self.by_epoch, self.save_optimizer = interval, by_epoch, save_optimizer
        self.out_dir, self.max_keep_ckpts, self.save_last = out_dir, max_keep_ckpts, save_last

    def before_run(self, runner):
        if not self.out_dir:
            self.out_dir = runner.work_dir

    def _save(self, runner, n, tmpl):
        if self.out_dir is None or not _is_rank0():
            return
        runner.save_checkpoint(self.out_dir, filename_tmpl=tmpl, save_optimizer=self.save_optimizer)
        if self.max_keep_ckpts > 0:
            for old in range(n - self.max_keep_ckpts * self.interval, 0, -self.interval):
                path = os.path.join(self.out_dir, tmpl.format(old))
                if not os.path.exists(path):
                    break
                os.remove(path)

    def after_train_iter(self, runner):
        if self.by_epoch or self.interval <= 0:
            return
        n = runner.iter + 1
        if n % self.interval == 0 or (self.save_last and runner.max_iters is not None and n == runner.max_iters):
            self._save(runner, n, 'iter_{}.pth')

    def after_train_epoch(self, runner):
        if not self.by_epoch or self.interval <= 0:
            return
        n = runner.epoch + 1
        last = getattr(runner, 'max_epochs', None)
        if n % self.interval == 0 or (self.save_last and last is not None and n == last):
            self._save(runner, n, 'epoch_{}.pth')


def _is_rank0():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class _PassThroughHook(Hook):
    def __init__(self, **kwargs):
        pass


for _n in ('WandbVisHook', 'TensorboardLoggerHook', 'WandbLoggerHook'):
    # logging back-ends are outside the hot path: registered so configs build
    HOOKS.register_module(type(_n, (_PassThroughHook,), {}))


@HOOKS.register_module()
class TextLoggerHook(Hook):
    """mmcv TextLoggerHook, the part a training run is watched through: every `interval` iterations rank 0 writes one line —
    iteration, learning rate(s), seconds per iteration over the window, peak device memory, and the mean of every logged
    value over the window (`log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook')])`).  The values stay on the
    device until here: one read-back per `interval` iterations.  File / json output and ETA are not reproduced."""

    def __init__(self, by_epoch=True, interval=10, ignore_last=True, reset_flag=False, interval_exp_name=1000, **kwargs):
        self.interval = int(interval)
        self._t = None
        self.lines = []            # what was written (rank 0), newest last; at most 100 kept

    def before_run(self, runner):
        self._t = time.time()

    def after_train_iter(self, runner):
        if (runner.iter + 1) % self.interval != 0:
            return
        runner._settle_lazy_logs()              # on EVERY rank: the collective behind the logged values (see there)
        if not _is_rank0():
            return
        now = time.time()
        dt = (now - (self._t or now)) / self.interval
        self._t = now
        parts = []
        lrs = getattr(runner.optimizer, 'optimizers', [runner.optimizer])
        parts.append('lr: ' + ' '.join('%.3e' % o.param_groups[0]['lr'] for o in lrs if getattr(o, 'param_groups', None)))
        parts.append('time: %.3f' % dt)
        if torch.cuda.is_available():
            parts.append('memory: %d' % (torch.cuda.max_memory_allocated() // (1 << 20)))
        buf = runner.log_buffer
        keys = [k for k in buf if buf[k]]
        if keys:
            def scalar(v, dev):
                v = v.detach() if torch.is_tensor(v) else torch.as_tensor(v)
                return v.float().reshape(()).to(dev)
            dev = next((v.device for k in keys for v in buf[k][-self.interval:] if torch.is_tensor(v) and v.is_cuda), 'cpu')
            means = torch.stack([torch.stack([scalar(v, dev) for v in buf[k][-self.interval:]]).mean() for k in keys])
            parts.extend('%s: %.4f' % (k, v) for k, v in zip(keys, means.tolist()))           # ONE read-back
        total = getattr(runner, 'max_iters', None)
        line = 'Iter [%d/%s]\t%s' % (runner.iter + 1, total if total is not None else '?', ', '.join(parts))
        self.lines = (self.lines + [line])[-100:]
        logger = getattr(runner, 'logger', None)
        if logger is not None and hasattr(logger, 'info'):
            logger.info(line)
        else:
            print(line, flush=True)


class IterLoader(object):
    """mmcv IterLoader: endless iterator over a loader, counting epochs."""

    def __init__(self, dataloader):
        self._dataloader = dataloader
        self.iter_loader = iter(dataloader)
        self._epoch = 0

    @property
    def epoch(self):
        return self._epoch

    def __next__(self):
        try:
            data = next(self.iter_loader)
        except StopIteration:
            self._epoch += 1
            if hasattr(getattr(self._dataloader, 'sampler', None), 'set_epoch'):
                self._dataloader.sampler.set_epoch(self._epoch)
            self.iter_loader = iter(self._dataloader)
            data = next(self.iter_loader)
        return data

    def __len__(self):
        return len(self._dataloader)


def _ensure_train_mode(model):
    """`model.train()` of the reference runners.  Walking the ~1300 sub-modules of the DetMatch model
    every iteration costs 3-6 ms of host time (hidden behind the previous step's device tail on this
    host, not on a slower or shared one), so a model may vouch for
    its own mode through `training_mode_ok()` (SSL does: student training, teacher eval, checked two
    levels deep); anything else gets the plain call."""
    inner = model.module if hasattr(model, 'module') else model
    ok = getattr(inner, 'training_mode_ok', None)
    if not (model.training and ok is not None and ok()):
        model.train()


class _RunnerBase(object):

    def __init__(self, model, optimizer=None, max_iters=None, work_dir=None, logger=None,
                 meta=None, batch_processor=None, **kwargs):
        self.model = model
        self.optimizer = optimizer
        self._max_iters = max_iters
        self.work_dir = work_dir
        self.logger = logger
        self._hooks = []
        self._iter = 0
        self._inner_iter = 0
        self._epoch = 0
        self.mode = None
        self.outputs = None
        self._lazy_logs = []
        self._log_buffer = dict()

    iter = property(lambda self: self._iter)
    epoch = property(lambda self: self._epoch)
    inner_iter = property(lambda self: self._inner_iter)
    max_iters = property(lambda self: self._max_iters)

    def register_hook(self, hook, late=False):
        """Hooks run in registration order; `late` ones (mmcv priority VERY_LOW: the loggers) stay behind every other hook,
        whenever those are registered (an EvalHook added after register_training_hooks still runs in front of the logger)."""
        if isinstance(hook, dict):
            hook = build_from_cfg(hook, HOOKS)
        n_late = getattr(self, '_n_late', 0)
        if late:
            self._hooks.append(hook)
            self._n_late = n_late + 1
        else:
            self._hooks.insert(len(self._hooks) - n_late, hook)

    # ---- checkpoints (mmcv BaseRunner.save_checkpoint / load_checkpoint / resume; mmcv format:
    # dict(meta=dict(epoch, iter, ...), state_dict=..., optimizer=...), 'module.' prefix stripped)
    def save_checkpoint(self, out_dir, filename_tmpl=None, save_optimizer=True, meta=None, create_symlink=True):
        by_iter = isinstance(self, IterBasedSSLRunner)
        if filename_tmpl is None:
            filename_tmpl = 'iter_{}.pth' if by_iter else 'epoch_{}.pth'
        n = self.iter + 1 if by_iter else self.epoch + 1
        info = dict(meta or {}, iter=self.iter + 1 if by_iter else self.iter, epoch=self.epoch + 1)
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, filename_tmpl.format(n))
        ckpt = dict(meta=info, state_dict={k: v.detach().cpu() for k, v in _inner(self.model).state_dict().items()})
        if save_optimizer and self.optimizer is not None:
            ckpt['optimizer'] = self.optimizer.state_dict()
        torch.save(ckpt, path)
        if create_symlink:
            link = os.path.join(out_dir, 'latest.pth')
            if os.path.lexists(link):
                os.remove(link)
            try:
                os.symlink(os.path.basename(path), link)
            except OSError:
                import shutil
                shutil.copy(path, link)
        return path

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        ckpt = torch.load(filename, map_location=map_location, weights_only=False)
        sd = ckpt.get('state_dict', ckpt)
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        _inner(self.model).load_state_dict(sd, strict=strict)
        from .. import dense_conv
        dense_conv.weights_changed()          # every packed copy of a convolution weight is stale now
        return ckpt

    def resume(self, checkpoint, resume_optimizer=True, map_location='cpu'):
        ckpt = self.load_checkpoint(checkpoint, map_location=map_location)
        self._epoch = ckpt['meta']['epoch']
        self._iter = ckpt['meta']['iter']
        self._inner_iter = 0
        if 'optimizer' in ckpt and resume_optimizer and self.optimizer is not None:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        return ckpt

    def register_training_hooks(self, lr_config=None, optimizer_config=None, custom_hooks=None,
                                momentum_config=None, checkpoint_config=None, log_config=None, **ignored):
        """Order = mmcv priorities for this set: LR (VERY_HIGH) < momentum (HIGH) < optimizer
        (ABOVE_NORMAL) < checkpoint (NORMAL) < custom (NORMAL) < loggers (VERY_LOW, `late`).  Of the logging back-ends of
        log_config only TextLoggerHook writes something (one line per interval); the others are pass-throughs."""
        if lr_config is not None:
            cfg = dict(lr_config)
            policy = cfg.pop('policy', 'step')
            assert policy in ('step', 'cyclic'), 'policies of the DetMatch recipes: step, cyclic'
            if policy == 'step':
                cfg.setdefault('by_epoch', isinstance(self, EpochBasedRunner))     # mmcv default: by_epoch=True
                self.register_hook(StepLrUpdaterHook(**cfg))
            else:
                self.register_hook(CyclicLrUpdaterHook(**cfg))
        if momentum_config is not None:
            cfg = dict(momentum_config)
            assert cfg.pop('policy') == 'cyclic'
            self.register_hook(CyclicMomentumUpdaterHook(**cfg))
        if optimizer_config is not None:
            self.register_hook(OptimizerHook(**dict(optimizer_config)))
        if checkpoint_config is not None:
            cfg = dict(checkpoint_config)
            cfg.setdefault('by_epoch', isinstance(self, EpochBasedRunner))
            self.register_hook(CheckpointHook(**cfg))
        for h in (custom_hooks or []):
            self.register_hook(h)
        if log_config is not None:
            for h in log_config.get('hooks', []):
                cfg = dict(h)
                cfg.setdefault('interval', log_config.get('interval', 10))
                self.register_hook(build_from_cfg(cfg, HOOKS), late=True)

    def call_hook(self, name):
        for h in self._hooks:
            getattr(h, name, lambda r: None)(self)

    def _after_step(self, outputs):
        if not isinstance(outputs, dict):
            raise TypeError('model.train_step() must return a dict')
        self.outputs = outputs
        lv = outputs.get('log_vars')
        if lv is not None and getattr(lv, '_pending', None) is not None:
            self._lazy_logs.append(lv)           # values not computed yet (LazyLogVars): buffered when somebody reads
        elif lv is not None:
            buf = self._log_buffer
            for k, v in lv.items():
                buf.setdefault(k, []).append(v)

    @staticmethod
    def _several_ranks():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _fold(self, lazy):
        for k, v in lazy.items():                # (first access computes the values: LazyLogVars._fill)
            self._log_buffer.setdefault(k, []).append(v)

    def _settle_lazy_logs(self):
        """With more than one rank the logged values are averaged by a collective when they are computed
        (base_detector.LazyLogVars._fill): every rank computes them HERE — the runner calls this behind the hooks of every
        iteration (i.e. behind the issued backward pass), a hook that wants the current iteration's values calls it first, on
        every rank (TextLoggerHook).  Reading `log_buffer` never starts that collective: a reader that only exists on rank 0
        (the printing half of a logger, bench.py) would start it alone, or in another place of the collective order."""
        if self._lazy_logs and self._several_ranks():
            pending, self._lazy_logs = self._lazy_logs, []
            for lv in pending:
                self._fold(lv)

    @property
    def log_buffer(self):
        """{key: [value per iteration]}; lazily parsed log_vars of finished iterations are folded in on access — with one
        rank.  With several ranks only `_settle_lazy_logs` folds them (see there)."""
        if self._lazy_logs and not self._several_ranks():
            pending, self._lazy_logs = self._lazy_logs, []
            for lv in pending:
                self._fold(lv)
        return self._log_buffer

    @log_buffer.setter
    def log_buffer(self, value):
        self._lazy_logs = []
        self._log_buffer = value


@RUNNERS.register_module()
class EpochBasedRunner(_RunnerBase):
    """mmcv EpochBasedRunner (1.3.16; the pre-training recipes pretrain_pvrcnn / pretrain_frcnn):
    run(data_loaders, workflow) sets max_iters = max_epochs * len(loader) and trains epoch by epoch,
    one model.train_step per batch."""

    def __init__(self, model, optimizer=None, max_epochs=None, max_iters=None, **kwargs):
        super().__init__(model, optimizer=optimizer, max_iters=max_iters, **kwargs)
        self._max_epochs = max_epochs

    max_epochs = property(lambda self: self._max_epochs)

    def train(self, data_loader, **kwargs):
        _ensure_train_mode(self.model)
        self.mode = 'train'
        self.data_loader = data_loader
        self._max_iters = self._max_epochs * len(data_loader)
        self.call_hook('before_epoch')
        for i, data_batch in enumerate(data_loader):
            self._inner_iter = i
            self.call_hook('before_train_iter')
            self._after_step(self.model.train_step(data_batch, self.optimizer, **kwargs))
            self.call_hook('after_train_iter')
            self._settle_lazy_logs()
            self._iter += 1
        self.call_hook('after_train_epoch')
        self.call_hook('after_epoch')
        self._epoch += 1

    def run(self, data_loaders, workflow=(('train', 1),), max_epochs=None, **kwargs):
        assert isinstance(data_loaders, list) and len(data_loaders) == len(workflow)
        if max_epochs is not None:
            self._max_epochs = max_epochs
        assert self._max_epochs is not None, 'max_epochs must be specified during instantiation'
        for i, (mode, epochs) in enumerate(workflow):
            if mode == 'train':
                self._max_iters = self._max_epochs * len(data_loaders[i])
                break
        self.call_hook('before_run')
        while self.epoch < self._max_epochs:
            for i, (mode, epochs) in enumerate(workflow):
                assert mode == 'train', 'only the train workflow is on the DetMatch path'
                for _ in range(epochs):
                    if self.epoch >= self._max_epochs:
                        break
                    self.train(data_loaders[i], **kwargs)
        self.call_hook('after_run')




@RUNNERS.register_module()
class IterBasedSSLRunner(_RunnerBase):
    """iter_based_ssl_runner.py:11-110: one labeled + one unlabeled batch per iteration, keys
    prefixed 'lab_' / 'unlab_'; data_batch['img_metas'] is the labeled metas (only used for
    num_samples)."""

    # Look-ahead (scheduling only; DM_LOOKAHEAD=0 disables): the batches of iteration i+1 are drawn before
    # iteration i is issued, and once i is issued the model prepares their weight-independent 3D geometry on a
    # side stream (SSL.prefetch_geometry) — the data-dependent size read-backs of the voxelizer / rulebooks
    # then overlap the tail of iteration i instead of idling the device at the step boundary.
    lookahead = os.environ.get('DM_LOOKAHEAD', '1') == '1'
    # draw_ahead: only the DRAW one iteration ahead (no geometry on a side stream): the batch's `ready` event then sits in
    # front of the previous iteration on the main stream, and the model's own early work can wait for it alone
    draw_ahead = False

    @staticmethod
    def _draw(lab_data_loader, unlab_data_loader):
        epoch = getattr(lab_data_loader, 'epoch', 0)       # as IterBasedRunner.train reads it: BEFORE this batch is drawn
        lab = next(lab_data_loader)
        unlab = next(unlab_data_loader)
        data_batch = {'lab_%s' % k: v for k, v in lab.items()}
        data_batch.update({'unlab_%s' % k: v for k, v in unlab.items()})
        data_batch['img_metas'] = lab['img_metas']
        ready = None
        if torch.cuda.is_available():
            ready = torch.cuda.Event()
            ready.record()
        return data_batch, ready, epoch

    def train(self, lab_data_loader, unlab_data_loader, **kwargs):
        _ensure_train_mode(self.model)
        self.mode = 'train'
        ahead = getattr(self, '_ahead', None)
        data_batch, ready, self._epoch = ahead if ahead is not None else self._draw(lab_data_loader, unlab_data_loader)
        self._ahead = None
        prefetch = getattr(_inner(self.model), 'prefetch_geometry', None)
        draw_ahead = self.lookahead or self.draw_ahead
        if draw_ahead and prefetch is not None and (self._max_iters is None or self.iter + 1 < self._max_iters):
            self._ahead = self._draw(lab_data_loader, unlab_data_loader)
        if ready is not None and prefetch is not None and ahead is not None:
            # the batch existed before the previous iteration was issued: the model may start what reads nothing but the
            # batch (geometry) and the EMA (teacher) without waiting for that iteration's tail (SSL._forward_train)
            _inner(self.model)._data_ready = ready
        self.call_hook('before_train_iter')
        # weight-gradient halves of the chained backward passes on the side stream for the length of this iteration
        # (chain.SIDE_WGRAD; scheduling only): needs the lanes and a gradient arena that is filled by collect(),
        # which waits for those kernels before it reads what they wrote
        from .. import chain as _chain
        inner = _inner(self.model)
        _chain.SIDE_WGRAD[0] = bool(getattr(inner, 'side_wgrad', False) and getattr(inner, 'two_lanes', False)
                                         and getattr(self.model, 'mode', None) == 'collect')
        try:
            self._after_step(self.model.train_step(data_batch, self.optimizer, **kwargs))
            self.call_hook('after_train_iter')
            self._settle_lazy_logs()
        finally:
            _chain.SIDE_WGRAD[0] = False
        if self._ahead is not None and self.lookahead:
            prefetch(self._ahead[0], self._ahead[1], tag='ahead%d_' % (self.iter & 1))
        self._inner_iter += 1
        self._iter += 1

    def run(self, data_loaders, workflow=(('train', 1),), max_iters=None, **kwargs):
        assert isinstance(data_loaders, list) and len(data_loaders) == 2  # labeled & unlabeled
        assert len(workflow) == 1 and workflow[0][0] == 'train'
        if max_iters is not None:
            self._max_iters = max_iters
        assert self._max_iters is not None, 'max_iters must be specified during instantiation'
        lab, unlab = IterLoader(data_loaders[0]), IterLoader(data_loaders[1])
        self.call_hook('before_run')
        self.call_hook('before_epoch')
        while self.iter < self._max_iters:
            self._inner_iter = 0
            for _ in range(workflow[0][1]):
                if self.iter >= self._max_iters:
                    break
                self.train(lab, unlab, **kwargs)
        self.call_hook('after_epoch')
        self.call_hook('after_run')
