"""SSL teacher-student detector and the 2D+3D MMDetector wrapper —
mmdet3d/models/detectors/ssl.py:20-380 and mmdetector.py:17-94.

Teacher and student share one parameter LAYOUT in two flat, 16-byte aligned arenas, so the
EMA of SSL._update_teacher (ssl.py:146-163: ~600 state-dict entries x 3 launches per
iteration in the reference) is one dm_ema_update_f32 launch plus one dm_ema_update_i64
launch for the integer buffers.
"""
import copy
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib
from .registry import DETECTORS, build_detector, build_ssl_module


def add_prefix(inputs, prefix):
    """mmseg.core.add_prefix"""
    return {'%s.%s' % (prefix, k): v for k, v in inputs.items()}


# Scheduling choices that were measured and kept (profiles/r05_ab_step_variants.txt, r06_ab_step_variants.txt): module
# attributes — the gradient-equality tests switch them (tests/test_ssl_gpu.py) — not environment switches.
_ISSUE_EARLY = True       # SSL modules' issue_early: unlabeled passes issued before their chain inputs exist
_EARLY_2D_BWD = True      # SSL._early_2d_backward: unlabeled 2D losses + deferred 2D trunk backward right after the last 2D module
_TRUNK_ON_2D_LANE = os.environ.get('DM_TRUNK_ON_2D_LANE', '0') == '1'     # the unlabeled 3D trunk on the 2D lane beside the supervised backward: measured WORSE (59.6 / 70.0 / 56.7 against
                              # 55.2 / 60.0 / 56.4 ms: it delays the teacher, whose read-back gates the glue; no gain with the final order either,
                              # profiles/r06_ab_step_variants.txt) — off, kept for the record
_TEACHER_AHEAD = os.environ.get('DM_TEACHER_AHEAD', '1') == '1'   # geometry + the teacher's 2D pass wait for the previous EMA (and the batch), not for the previous iteration's last backward
_TEACHER_AHEAD_ALL = os.environ.get('DM_TEACHER_AHEAD_ALL', '0') == '1'   # ... the teacher's 3D pass too (else: its 2D pass only)
_SUP_BWD_PER_LANE = os.environ.get('DM_SUP_BWD_PER_LANE', '0') == '1'   # supervised 2D losses back-propagated on the 2D lane
_2D_INSIDE_3D = True      # the last 2D module between the issue and the read-back of its 3D neighbour: 60.2-60.5 against 61.4-62.0 ms (profiles/r06_ab_step_variants.txt)


class _LaneDict(dict):
    """Batch-dict shell that knows on which HIP stream ("lane") each entry was produced.  Reading
    an entry produced on the other lane makes the reading stream wait for the producer's event and
    registers the tensors with the reading stream (caching-allocator safety) — the data-flow edges
    between the 2D and the 3D branch become stream dependencies, everything else runs concurrently."""

    def __init__(self, data, lanes):
        super().__init__(data)
        self._lanes = lanes
        self._event = {}          # key -> (lane id, event or None while the producer still runs)

    # loss / log dictionaries are merged entry by entry without touching the existing tensors
    # (SSL._sum_update_losses); they are made visible to the main lane by _Lanes.join()
    UNTRACKED = ('sup_losses', 'ssl_losses', 'vis', 'log_vars')

    def __setitem__(self, k, v):
        ln = self._lanes
        if k in self.UNTRACKED:
            return super().__setitem__(k, v)
        if isinstance(v, dict) and not isinstance(v, _LaneDict):
            v = _LaneDict(v, ln)
        super().__setitem__(k, v)
        tok = [ln.current, None]
        self._event[k] = tok
        ln.pending.append(tok)

    def _sync(self, k):
        tok = self._event.get(k)
        ln = self._lanes
        if tok is not None and tok[0] != ln.current:
            if tok[1] is None:
                # a producer that is still being issued: never with the modules run one after the other
                raise RuntimeError('batch-dict entry %r is read on lane %d while lane %d is still '
                                   'producing it' % (k, ln.current, tok[0]))
            cur = torch.cuda.current_stream()
            cur.wait_event(tok[1])
            ln.record(super().get(k), cur)

    def __getitem__(self, k):
        self._sync(k)
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._sync(k)
        return super().get(k, default)


class _RngWindows(object):
    """Random numbers per SSL module, independent of the order the modules are issued in: module i of the
    iteration (labeled chain, then unlabeled chain) draws from the device generator's Philox stream at
    offset base + (i + 1) * 2^32, and the iteration leaves the generator at base + (n + 1) * 2^32.  The
    serial order, the hoisted teacher and a 2D pass issued from inside its 3D partner all see the same
    sampling keys and dropout masks."""

    WINDOW = 1 << 32

    def __init__(self, device, modules):
        self.gen = None
        if device.type == 'cuda':
            idx = device.index if device.index is not None else torch.cuda.current_device()
            self.gen = torch.cuda.default_generators[idx]
        self.index = {id(m): i for i, m in enumerate(modules)}
        self.base = self.gen.get_offset() if self.gen is not None else 0

    def enter(self, module):
        i = self.index.get(id(module))
        if self.gen is not None and i is not None:
            self.gen.set_offset(self.base + (i + 1) * self.WINDOW)

    def offset(self):
        return self.gen.get_offset() if self.gen is not None else None

    def restore(self, off):
        if self.gen is not None:
            self.gen.set_offset(off)

    def finish(self):
        if self.gen is not None:
            self.gen.set_offset(self.base + (len(self.index) + 1) * self.WINDOW)


class _Lanes(object):
    """HIP streams for SSL.forward_train.  Lane 0 = the caller's stream: the student's 3D passes
    (stateful modules, strictly ordered) and the final aggregation; lane 1: the modules that run a 2D
    detector; lane 2: the teacher's 3D pass and all the light glue modules (filters, transforms,
    Hungarian matching, consistency) — so the pseudo-label chain, including its host read-backs,
    never queues behind the student's (early) backward on lane 0."""

    N = 3
    _side = {}

    def __init__(self, device, mode='branches'):
        # mode 'branches': the three lanes above.  mode 'glue': TWO lanes — every detector pass (student
        # and teacher, 2D and 3D) stays on the caller's stream, strictly ordered, so the heavy kernels
        # never share the device; only the light pseudo-label glue (filters, transforms, matching, with
        # all its host read-backs) runs on a side stream, underneath the supervised passes.
        self.mode = mode
        self.main = torch.cuda.current_stream(device)
        from .. import _lib
        _lib.MAIN_STREAM[0] = self.main.cuda_stream      # vendor GEMMs stay on this lane (_lib.blas_linear)
        pool = _Lanes._side.get(device.index)
        if pool is None:
            # (a high-priority stream for the teacher-3D + glue lane measured no different: DESIGN.md §9)
            pool = _Lanes._side[device.index] = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
        self.streams = [self.main] + pool
        self.current = 0
        self.pending = []
        self.partial = {}             # lane -> event the main lane waits for in join() instead of the whole lane
        self.rng = None               # _RngWindows of the iteration

    def stream(self, lane):
        return self.streams[lane]

    def lane_of(self, module):
        attr = getattr(module, 'ssl_obj_attr', None)
        if self.mode == 'glue':
            return 1 if attr is None else 0
        if attr is None:
            return 2                       # glue
        if attr.endswith('detector_2d'):
            return 1
        return 0 if attr.startswith('student') else 2

    @staticmethod
    def record(value, stream):
        if isinstance(value, torch.Tensor):
            if value.is_cuda:
                value.record_stream(stream)
        elif hasattr(value, 'tensor') and isinstance(value.tensor, torch.Tensor):
            _Lanes.record(value.tensor, stream)
        elif hasattr(value, 'full') and hasattr(value, 'keep'):      # ssl_modules.Masked: a lazily filtered entry
            _Lanes.record(value.full, stream)
            _Lanes.record(value.keep, stream)
        elif isinstance(value, (list, tuple)):
            for v in value:
                _Lanes.record(v, stream)
        elif isinstance(value, dict):
            for v in value.values():
                _Lanes.record(v, stream)

    def fork(self):
        """The inputs (and the weights, just updated by optimizer / EMA) live on the main stream."""
        for s in self.streams[1:]:
            s.wait_stream(self.main)

    def run(self, module, ssl_obj, batch_dict, method='forward', lane=None):
        lane = self.lane_of(module) if lane is None else lane
        self.current, self.pending = lane, []
        with torch.cuda.stream(self.stream(lane)):
            if self.rng is not None:
                self.rng.enter(module)
            if not getattr(module, 'takes_masked', False):
                compact_masked(batch_dict)
            out = getattr(module, method)(ssl_obj, batch_dict)
            ev = torch.cuda.Event()
            ev.record(self.stream(lane))
        for tok in self.pending:
            tok[1] = ev
        self.current, self.pending = 0, []
        return out

    def join(self, *values):
        """Everything issued on the other lanes so far becomes visible to the main lane."""
        for i, s in enumerate(self.streams[1:], 1):
            ev = self.partial.get(i)
            if ev is not None:
                self.main.wait_event(ev)
            else:
                self.main.wait_stream(s)
        for v in values:
            self.record(v, self.main)


def compact_masked(d):
    """Box lists as the reference's modules hand them to each other: every lazily filtered entry (ssl_modules.Masked)
    of the (nested) batch dict compacted in place.  Run in front of a module that does not declare `takes_masked` — a
    user's own SSL module sees plain (boxes, scores) tuples, whatever the built-in filters did before it."""
    for k in list(d.keys()):
        v = dict.__getitem__(d, k)             # (a look, not a read: no lane dependency for entries that hold no mask)
        if isinstance(v, dict):
            compact_masked(v)
        elif isinstance(v, list) and any(hasattr(e, 'full') and hasattr(e, 'keep') for e in v):
            v = d[k]                           # a read on this lane: waits for the producer's event
            d[k] = [e.entry() if (hasattr(e, 'full') and hasattr(e, 'keep')) else e for e in v]


class _Arena(object):
    """Re-homes every tensor of a state_dict in two flat buffers (fp32 / int64).  `first_keys`
    (state-dict keys) are laid out first, in that order — the student's trainable parameters in the
    order of the gradient arena, so optimizer, gradient and EMA kernels share one layout."""

    def __init__(self, module, first_keys=None):
        sd = module.state_dict(keep_vars=True)
        self.keys = list(sd.keys())
        other = [k for k, v in sd.items() if v.dtype not in (torch.float32, torch.int64)]
        assert not other, 'unsupported state dtypes: %s' % other
        first_keys = list(first_keys or [])
        seen = set(first_keys)
        self.f_keys = first_keys + [k for k, v in sd.items()
                                    if v.dtype == torch.float32 and k not in seen]
        self.i_keys = [k for k, v in sd.items() if v.dtype == torch.int64]
        f_items = [(k, sd[k]) for k in self.f_keys]
        i_items = [(k, sd[k]) for k in self.i_keys]
        dev = f_items[0][1].device if f_items else torch.device('cpu')
        # each tensor starts on a 4-element boundary so the float4 kernels never straddle
        pad = lambda n: (n + 3) // 4 * 4
        self.flat_f = torch.zeros(sum(pad(v.numel()) for _, v in f_items), dtype=torch.float32, device=dev)
        self.flat_i = torch.zeros(sum(v.numel() for _, v in i_items), dtype=torch.int64, device=dev)
        self.f_offset = {}
        off = 0
        with torch.no_grad():
            for k, v in f_items:
                n = v.numel()
                view = self.flat_f[off:off + n].view(v.shape)
                view.copy_(v.data)
                v.data = view
                self.f_offset[k] = off
                off += pad(n)
            off = 0
            for k, v in i_items:
                n = v.numel()
                view = self.flat_i[off:off + n].view(v.shape)
                view.copy_(v.data)
                v.data = view
                off += n


@DETECTORS.register_module()
class MMDetector(nn.Module):
    """mmdetector.py:17-94: holds detector_2d and detector_3d, used independently."""

    def __init__(self, detector_2d, detector_3d, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        detector_2d = dict(detector_2d)
        detector_3d = dict(detector_3d)
        detector_2d['train_cfg'] = train_cfg['detector_2d'] if train_cfg is not None else None
        detector_2d['test_cfg'] = test_cfg['detector_2d']
        self.detector_2d = build_detector(detector_2d)
        detector_3d['train_cfg'] = train_cfg['detector_3d'] if train_cfg is not None else None
        detector_3d['test_cfg'] = test_cfg['detector_3d']
        self.detector_3d = build_detector(detector_3d)
        self._load_pretrained(pretrained)

    def _load_pretrained(self, pretrained):
        if pretrained is not None:
            assert isinstance(pretrained, dict)
            self.detector_2d.load_state_dict(
                torch.load(pretrained['detector_2d'], map_location='cpu')['state_dict'])
            self.detector_3d.load_state_dict(
                torch.load(pretrained['detector_3d'], map_location='cpu')['state_dict'])

    def simple_test(self, **kwargs):
        do_2d = 'img' in kwargs
        points = kwargs.pop('points')
        if do_2d:
            results_2d = self.detector_2d.simple_test(**kwargs)
            kwargs.pop('img')
        kwargs['points'] = points
        results_3d = self.detector_3d.simple_test(**kwargs)
        if do_2d:
            return [dict(results_2d=r2, results_3d=r3) for r2, r3 in zip(results_2d, results_3d)]
        return results_3d


@DETECTORS.register_module()
class SSL(nn.Module):

    def __init__(self, model_cfg, ssl_cfg, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        if self.train_cfg is not None:
            self.ema_params = self.train_cfg['ssl']['ema_params']
            self.ema_decay = self.ema_params['ema_decay']
            self.true_avg_rampup = self.ema_params.get('true_avg_rampup', False)
            self.rampup_start_decay = self.ema_params.get('rampup_start_decay', 0.5)
            self.use_student_bn_stats_for_teacher = self.ema_params.get(
                'use_student_bn_stats_for_teacher', False)
            self.ssl_weight_params = self.train_cfg['ssl']['weight_params']
            self.ssl_weight = self.ssl_weight_params['weight']
            self.ssl_weight_rampup_start_iter = self.ssl_weight_params.get(
                'weight_rampup_start_iter', 0)
            self.ssl_weight_rampup_num_iter = self.ssl_weight_params.get('weight_rampup_num_iter', 0)
            self.set_teacher_eval = self.train_cfg['ssl'].get('set_teacher_eval', False)
            self.epoch, self.iter = None, None   # filled in by ModelIterEpochHook
        teacher_cfg = copy.deepcopy(dict(model_cfg))
        teacher_cfg['train_cfg'] = train_cfg['teacher'] if train_cfg is not None else None
        teacher_cfg['test_cfg'] = test_cfg['teacher']
        teacher_cfg['pretrained'] = pretrained
        self.teacher = build_detector(teacher_cfg)
        for p in self.teacher.parameters():
            p.requires_grad = False
        student_cfg = copy.deepcopy(dict(model_cfg))
        student_cfg['train_cfg'] = train_cfg['student'] if train_cfg is not None else None
        student_cfg['test_cfg'] = test_cfg['student']
        student_cfg['pretrained'] = pretrained
        self.student = build_detector(student_cfg)
        ssl_cfg = dict(ssl_cfg)
        self.lab_ssl_modules, self.unlab_ssl_modules = [], []
        for name, target in (('labeled', self.lab_ssl_modules), ('unlabeled', self.unlab_ssl_modules)):
            if name in ssl_cfg:
                lst = ssl_cfg[name] if isinstance(ssl_cfg[name], (list, tuple)) else [ssl_cfg[name]]
                target.extend(build_ssl_module(c) for c in lst)
        self._arenas = None

    # ---- state-dict fan-out (ssl.py:102-127) ------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        self._ema_done = None          # the teacher is rewritten on the caller's stream: no early teacher pass in the next iteration
        loading_pretrained = not any('teacher' in k for k in state_dict.keys())
        if loading_pretrained:
            self.teacher.load_state_dict(state_dict)
            self.student.load_state_dict(copy.deepcopy(state_dict))
            for k in list(state_dict.keys()):
                v = state_dict.pop(k)
                state_dict['teacher.' + k] = v
                state_dict['student.' + k] = copy.deepcopy(v)
        else:
            super()._load_from_state_dict(state_dict, prefix, local_metadata, strict,
                                          missing_keys, unexpected_keys, error_msgs)

    # ---- EMA ------------------------------------------------------------------------
    def _get_curr_ema_decay(self):
        """ssl.py:129-144"""
        if self.ema_params.get('true_avg_rampup', False):
            ema_start_iter = max(round(1 / (1 - self.rampup_start_decay)), 2)
            return min(1 - 1 / (self.iter + ema_start_iter), self.ema_params['ema_decay'])
        return self.ema_params['ema_decay']

    def build_arenas(self, ddp=None):
        """Call once the model sits on its device (and after any checkpoint load).  With a
        FlatGradDDP the student's trainable parameters come first, in gradient-arena order, and
        `ddp.flat_params` becomes that prefix of the student arena."""
        self._frozen_bn_differs = None
        self._ema_done = None          # (the tensors are re-homed on the caller's stream)
        first = None
        if ddp is not None:
            name_of = {id(p): n for n, p in self.student.named_parameters()}
            first = [name_of[id(p)] for p in ddp.order]
        self._arenas = (_Arena(self.teacher, first), _Arena(self.student, first))
        t, s = self._arenas
        assert t.f_keys == s.f_keys and t.i_keys == s.i_keys, 'teacher / student state dicts differ'
        if ddp is not None:
            ddp.flat_params = s.flat_f[:ddp.flat.numel()]
            assert ddp.check_param_arena(), 'student arena prefix does not match the gradient arena'
        return self._arenas

    def _update_teacher(self):
        """ssl.py:146-163: new_t = s * (1 - d) + t * d for EVERY state-dict entry (BN running
        stats and integer buffers included)."""
        d = float(self._get_curr_ema_decay())
        if self._arenas is None:
            self.build_arenas()
        t, s = self._arenas
        L = _lib.lib()
        if t.flat_f.is_cuda:
            _lib.check(L.dm_ema_update_f32(_lib.ptr(t.flat_f), _lib.ptr(s.flat_f), t.flat_f.numel(),
                                           d, _lib.stream()), 'dm_ema_update_f32')
            _lib.check(L.dm_ema_update_i64(_lib.ptr(t.flat_i), _lib.ptr(s.flat_i), t.flat_i.numel(),
                                           d, _lib.stream()), 'dm_ema_update_i64')
        else:
            raise _lib.DetMatchHipError('EMA runs on the MI355X only')
        self._bump_frozen_bn()
        from .. import dense_conv
        # the teacher's conv weights were rewritten through raw pointers (the student's were not)
        dense_conv.weights_changed(t.flat_f.data_ptr(), t.flat_f.data_ptr() + 4 * t.flat_f.numel())
        if self.use_student_bn_stats_for_teacher:
            tsd, ssd = self.teacher.state_dict(), self.student.state_dict()
            for k in tsd:
                if 'running' in k:
                    tsd[k].copy_(ssd[k])

    def _bump_frozen_bn(self):
        """The EMA kernel rewrites the teacher's tensors through raw pointers.  Frozen-BatchNorm
        layers cache their affine map; it only needs refreshing if teacher and student constants
        differ (then the EMA really moves them) — checked once."""
        differs = getattr(self, '_frozen_bn_differs', None)
        if differs is None:
            from ..mm2d.backbone import FrozenBN
            differs = False
            for mt, ms in zip(self.teacher.modules(), self.student.modules()):
                if isinstance(mt, FrozenBN):
                    for a, b in ((mt.weight, ms.weight), (mt.bias, ms.bias),
                                 (mt.running_mean, ms.running_mean), (mt.running_var, ms.running_var)):
                        differs = differs or not torch.equal(a, b)
            self._frozen_bn_differs = differs
        if differs:
            from ..mm2d.backbone import FrozenBN
            FrozenBN.GENERATION += 1

    # ---- loss bookkeeping -------------------------------------------------------------
    def _get_curr_ssl_weight(self):
        """ssl.py:165-181"""
        if self.ssl_weight_rampup_num_iter == 0:
            return self.ssl_weight
        if self.iter < self.ssl_weight_rampup_start_iter:
            return 0.0
        current = np.clip(self.iter - self.ssl_weight_rampup_start_iter, 0,
                          self.ssl_weight_rampup_num_iter)
        phase = 1.0 - current / self.ssl_weight_rampup_num_iter
        return self.ssl_weight * np.exp(-5.0 * phase * phase)

    def _collapse_losses(self, losses):
        for name, value in list(losses.items()):
            if isinstance(value, torch.Tensor):
                # mean() of a 0-d tensor is the identity: skipped so that merging loss dicts never
                # touches entries another lane (HIP stream) may still be computing
                losses[name] = value.mean() if value.dim() > 0 else value
            elif isinstance(value, list):
                losses[name] = sum(_l.mean() for _l in value)
            else:
                raise TypeError('%s is not a tensor or list of tensors' % name)
        return losses

    def _sum_update_losses(self, losses, new_losses):
        losses = self._collapse_losses(losses)
        new_losses = self._collapse_losses(new_losses)
        for k in new_losses:
            if k in losses:
                lanes = getattr(self, '_lanes', None)
                if lanes is not None:      # rare: the same key from two lanes
                    cur = torch.cuda.current_stream()
                    for st in lanes.streams:
                        if st != cur:
                            cur.wait_stream(st)
                    lanes.record(losses[k], cur)
                losses[k] = losses[k] + new_losses[k]
            else:
                losses[k] = new_losses[k]
        return losses

    def _parse_losses(self, losses):
        """mmdet BaseDetector._parse_losses: loss = sum of the entries whose key contains 'loss'.  The reference
        all-reduces every logged scalar separately (C2 in SURVEY §2.2); here the loss comes first and alone, and the
        logged values travel in ONE packed all-reduce when somebody reads them (base_detector.LazyLogVars) — this
        sits between the last glue module and the last backward pass, with the main lane idle."""
        from .base_detector import DetectorStepMixin
        return DetectorStepMixin._parse_losses(self, losses)

    # ---- step ------------------------------------------------------------------------------
    @staticmethod
    def _subtree_mode(module, training):
        """True when `module`, its children and grandchildren all have `.training == training` (the modes
        are only ever switched by whole-subtree train() / eval() calls)."""
        for m in (module,) + tuple(module.children()):
            if m.training != training or any(c.training != training for c in m.children()):
                return False
        return True

    def training_mode_ok(self):
        """The state every training iteration starts from after the first one: student in training
        mode, teacher in eval mode (set_teacher_eval) — lets the runner skip a redundant model.train()."""
        return self.training and self._subtree_mode(self.student, True) and \
            (self._subtree_mode(self.teacher, False) if self.set_teacher_eval else self._subtree_mode(self.teacher, True))

    def train_step(self, data, optimizer=None):
        """ssl.py:214-253"""
        if self.set_teacher_eval and not self._subtree_mode(self.teacher, False):
            self.teacher.eval()
        losses = self(**data)
        vis = losses.pop('vis', dict())
        log_vars = losses.pop('log_vars', dict())
        loss, log_vars_ = self._parse_losses(losses)
        # (ssl.py:249: log_vars.update(log_vars_) — the parsed values win; kept lazy: base_detector.LazyLogVars)
        if log_vars:
            merged = dict(log_vars)
            log_vars_.update({k: v for k, v in merged.items() if k not in losses})
        return dict(loss=loss, log_vars=log_vars_, num_samples=len(data['img_metas']), vis=vis)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            kwargs.pop('img_metas', None)
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def _share_2d_trunk(self, lab_dict, unlab_dict, lane_mode):
        """Scheduling only (FasterRCNN.prefetch_trunk): one backbone + FPN + RPN pass of a 2D detector for ALL
        the image batches its training modules of this iteration will be called with (the student's labeled and
        unlabeled images).  Needs a driver that calls finish_deferred_backward() after the last backward pass
        (OptimizerHook does): `share_2d_trunk` is set by the callers that have one."""
        self._deferred = []
        if not getattr(self, 'share_2d_trunk', False) or lane_mode not in (None, 'glue', 'serial', 'branches') \
                or not torch.is_grad_enabled():
            return
        from .bbox_utils import mlvl_get, mlvl_getattr
        from .ssl_modules import HardPseudoLabel_2D, TwoStageSupervised_2D
        groups = {}
        for chain, d in ((self.lab_ssl_modules, lab_dict), (self.unlab_ssl_modules, unlab_dict)):
            for m in chain:
                if isinstance(m, TwoStageSupervised_2D):
                    img = mlvl_get(d, m.batch_dict_key + '.img')
                elif isinstance(m, HardPseudoLabel_2D):
                    img = mlvl_get(d, m.target_img_key)
                else:
                    continue
                det = mlvl_getattr(self, m.ssl_obj_attr)
                if torch.is_tensor(img) and hasattr(det, 'prefetch_trunk'):
                    groups.setdefault(id(det), (det, []))[1].append(img)
        lanes = getattr(self, '_lanes', None)
        for det, imgs in groups.values():
            if len(imgs) < 2:
                continue
            if lanes is not None and lane_mode == 'branches':
                # the shared pass belongs to the 2D lane: the modules that consume its slices run there, and autograd
                # replays its (deferred) backward on the stream of its forward — underneath the 3D backward
                side = lanes.stream(1)
                lanes.current = 1
                try:
                    with torch.cuda.stream(side):
                        ok = det.prefetch_trunk(imgs)
                finally:
                    lanes.current = 0
            else:
                ok = det.prefetch_trunk(imgs)
            if ok:
                self._deferred.append(det)

    def finish_deferred_backward(self):
        for det in getattr(self, '_deferred', []):
            det.finish_deferred_backward()
        self._deferred = []
        ev = self.__dict__.pop('_trunk_bwd_done', None)
        if ev is not None:      # the trunk backward ran early on the 2D lane: whoever reads its gradients next waits for it
            torch.cuda.current_stream().wait_event(ev)

    def _early_2d_backward(self, lanes, module, d, ssl_weight):
        """'branches' only.  The student's 2D trunk receives gradient from the supervised 2D module (back-propagated
        with the labeled chain) and from this one — the consistency loss reaches the 3D student only
        (configs/detmatch/001/detmatch/split_0.py: its 2D side is the detached teacher).  So once this module has
        run, its losses AND the deferred trunk backward can be issued on the 2D lane, underneath the rest of the
        unlabeled chain, instead of behind the whole 3D backward at the end of the iteration (the 2D lane idled there
        while the host issued the 3D backward: 26 ms of device time after the last forward kernel).  d(sum) = sum of d."""
        side = lanes.stream(1)
        before = dict(d['ssl_losses'])
        d = lanes.run(module, self, d)
        cur = self._collapse_losses(d['ssl_losses'])
        picked = [(k, v) for k, v in cur.items() if (k not in before or before[k] is not v) and v.requires_grad]
        if any(k in before for k, _ in picked):
            return d
        w = float(ssl_weight)
        with torch.cuda.stream(side):
            terms = []
            for k, v in picked:
                if 'loss' in k:
                    terms.append(v if (w == 1.0 or '.metrics' in k or '.acc' in k) else v * w)
                cur[k] = v.detach()
            d['ssl_losses'] = cur
            pre = torch.cuda.Event()
            if terms:
                sum(terms).backward()
            pre.record(side)
            self.finish_deferred_backward()
            done = torch.cuda.Event()
            done.record(side)
        lanes.partial[1] = pre            # join(): the main lane needs the 2D lane up to here only ...
        self._trunk_bwd_done = done       # ... the trunk's gradients are waited for where they are read
        return d

    def prefetch_geometry(self, data, ready=None, tag='ahead'):
        """The weight-independent geometry (voxels, rulebooks, FPS key points) of every 3D pass of a FUTURE
        iteration's batch `data` (the dict train_step will be called with), on a side stream: its size
        read-backs then wait for a few small kernels instead of for the step that is still executing, and
        the iteration itself starts with the geometry on the table (OpenPCDetDetector._geom_cache).  `ready`:
        event after which the batch's tensors exist.  Scheduling only — geometry depends on the points."""
        lab_stu, unlab_stu = data.get('lab_stu'), data.get('unlab_stu')
        if lab_stu is None or isinstance(unlab_stu, list):
            return
        dev = next(self.student.parameters()).device
        if dev.type != 'cuda':
            return
        if getattr(self, '_geom_stream', None) is None:
            self._geom_stream = _lib.aux_stream(dev)      # shared with the key-point FPS (see _lib.aux_stream)
        dicts = (dict(stu=lab_stu, tea=data.get('lab_tea')), dict(stu=unlab_stu, tea=data.get('unlab_tea')))
        if ready is not None:
            self._geom_stream.wait_event(ready)
        with torch.cuda.stream(self._geom_stream):
            jobs = []
            for chain, d in zip((self.lab_ssl_modules, self.unlab_ssl_modules), dicts):
                for m in chain:
                    if hasattr(m, 'prefetch_steps'):
                        steps = m.prefetch_steps(self, d, '%s%d' % (tag, len(jobs)))
                        if steps is not None:
                            jobs.append(steps)
            if jobs:
                from ..spconv.ops import drive_steps_together
                drive_steps_together(jobs)
        # the batch's point clouds were allocated on another stream and are read here: keep the allocator informed
        for d in dicts:
            for part in d.values():
                for t in (part or {}).get('points', []) if isinstance(part, dict) else []:
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(self._geom_stream)

    def _run_and_backprop(self, run, module, d, early, ssl_weight):
        """Run one SSL module; when it is flagged `self_contained_losses` (its forward adds losses and
        nothing else) back-propagate the new loss terms right away, with the weight they carry in the
        final sum, and keep their detached values.  Scheduling only (d(sum) = sum of d): the
        device-bound backward of that pass then runs underneath the host-bound modules that follow,
        and its graph is freed early.  Gradients accumulate as for any second backward pass."""
        if not (early and getattr(module, 'self_contained_losses', False)):
            return run(module, d)
        before = {k: dict(d[k]) for k in ('sup_losses', 'ssl_losses') if k in d}
        d = run(module, d)
        # first pass: collect (group, key, value, weight) without touching anything — a key that two modules
        # add to only exists as its sum, so nothing of this module may be detached then
        picked, collapsed = [], {}
        for group, old in before.items():
            w = 1.0 if group == 'sup_losses' else float(ssl_weight)
            cur = collapsed[group] = self._collapse_losses(d[group])
            for k, v in cur.items():
                if (k not in old or old[k] is not v) and v.requires_grad:
                    if k in old:
                        return d
                    picked.append((group, k, v, w))
        terms = []
        for group, k, v, w in picked:
            if 'loss' in k:
                terms.append(v if (w == 1.0 or '.metrics' in k or '.acc' in k) else v * w)
            collapsed[group][k] = v.detach()
        for group, cur in collapsed.items():
            d[group] = cur
        if terms:
            from ..spconv.ops import deferred_weight_grads
            with deferred_weight_grads():      # the pass's sparse weight gradients in one launch pair
                sum(terms).backward()
            hook = getattr(self, 'after_partial_backward', None)
            if hook is not None:      # e.g. FlatGradDDP.collect: batched adds into the flat arena
                hook()
        return d

    def _sup_backward_on_lane(self, lanes, module, d, done):
        """'branches' only.  A supervised 2D module's losses are back-propagated on the 2D lane right behind its forward
        (heads only: the shared trunk's backward is deferred, _share_2d_trunk) instead of together with the 3D losses
        on the main lane — whose supervised backward then starts when the 3D forward is done, not when the 2D lane
        (teacher pass + shared trunk + this module: 19 ms) has caught up.  d(sum) = sum of d; the gradients are disjoint
        from the 3D detector's.  `done`: gets the event behind the lane's backward (whoever reads gradients waits)."""
        before = dict(d['sup_losses'])
        d = lanes.run(module, self, d)
        cur = d['sup_losses']
        new = [(k, v) for k, v in cur.items() if k not in before or before[k] is not v]
        if any(k in before for k, _ in new):
            return d                   # summed into another module's key: left to the joint backward
        side = lanes.stream(lanes.lane_of(module))
        with torch.cuda.stream(side):
            terms = []
            for k, v in new:
                v = self._collapse_losses({k: v})[k]
                if 'loss' in k and v.requires_grad:
                    terms.append(v)
                cur[k] = v.detach()
            if terms:
                sum(terms).backward()
                ev = torch.cuda.Event()
                ev.record(side)
                done.append(ev)
        return d

    def _issue_geometry(self, lab_dict, unlab_dict):
        """The weight-independent geometry of every pass of the iteration, issued up front on the current stream."""
        jobs = []
        for chain, d in ((self.lab_ssl_modules, lab_dict), (self.unlab_ssl_modules, unlab_dict)):
            for m in chain:
                steps = None
                if hasattr(m, 'prefetch_steps') and getattr(m.prefetch, '__func__', None) is \
                        getattr(type(m), 'prefetch', None):
                    steps = m.prefetch_steps(self, d, 'rulebook%d' % len(jobs))
                if steps is not None:
                    jobs.append(steps)
                elif hasattr(m, 'prefetch'):
                    m.prefetch(self, d)
        if jobs:
            # the passes' size read-backs (voxel count, N_out of the four strided rulebooks) in lockstep:
            # one device->host copy per round for all of them; their key-point FPS as ONE launch behind the rulebooks
            from ..pcdet.pfe import FpsBatch
            from ..spconv.ops import drive_steps_together
            with FpsBatch():
                drive_steps_together(jobs)

    def forward_train(self, lab_stu, lab_tea, unlab_stu, unlab_tea, *args, **kwargs):
        """ssl.py:255-350"""
        from ..bn_relu import deferred_counters
        with deferred_counters():      # BatchNorm call counters: one multi-tensor add before the EMA reads them
            return self._forward_train(lab_stu, lab_tea, unlab_stu, unlab_tea, *args, **kwargs)

    def _forward_train(self, lab_stu, lab_tea, unlab_stu, unlab_tea, *args, **kwargs):
        if isinstance(unlab_stu, list):
            unlab_stu = self._collate(unlab_stu)
            unlab_tea = self._collate(unlab_tea)
        assert lab_stu.get('gt_bboxes_ignore', None) is None and \
            lab_tea.get('gt_bboxes_ignore', None) is None
        lab_dict = dict(stu=lab_stu, tea=lab_tea, sup_losses=dict(), ssl_losses=dict())
        unlab_dict = dict(stu=unlab_stu, tea=unlab_tea, ssl_losses=dict())
        lanes = None
        dev = next(self.student.parameters()).device
        lane_mode = getattr(self, 'lane_mode', None) or ('branches' if getattr(self, 'two_lanes', False) else None)
        if lane_mode and dev.type == 'cuda':
            # modules on several HIP streams; data-flow edges become event waits (_LaneDict)
            lanes = _Lanes(dev, lane_mode)
            lab_dict = _LaneDict({k: (_LaneDict(v, lanes) if isinstance(v, dict) else v)
                                  for k, v in lab_dict.items()}, lanes)
            unlab_dict = _LaneDict({k: (_LaneDict(v, lanes) if isinstance(v, dict) else v)
                                    for k, v in unlab_dict.items()}, lanes)
        self._lanes = lanes
        rng = _RngWindows(dev, list(self.lab_ssl_modules) + list(self.unlab_ssl_modules))
        if lanes is not None:
            lanes.rng = rng

        def run_serial(m, d):
            rng.enter(m)
            if not getattr(m, 'takes_masked', False):
                compact_masked(d)
            return m.forward(self, d)

        run = (lambda m, d: lanes.run(m, self, d)) if lanes is not None else run_serial

        branches_first = lanes is not None and lanes.mode == 'branches' and _ISSUE_EARLY
        ahead = None
        ready = self.__dict__.pop('_data_ready', None)      # (set by the runner for THIS iteration only)
        if branches_first and _TEACHER_AHEAD:
            ema = getattr(self, '_ema_done', None)
            if ema is not None and ready is not None:
                ahead = (ema, ready)
        if ahead is not None:
            # Scheduling only ('branches').  Of the previous iteration, the teacher reads the EMA only — which ran BEFORE
            # that iteration's last backward pass (ssl.py:348) — and the geometry reads nothing but the batch.  So the 2D
            # and the teacher lane do not wait for the main stream's tail (backward, clip, optimizer: 10+ ms of device
            # time behind the host) but for the EMA's event and the batch's: the geometry of all passes (+ the key-point
            # FPS) goes to the teacher lane, its ONE size read-back returns at once, and the teacher's 2D pass is issued
            # and runs underneath the student's last backward.  (The teacher's 3D pass stays in chain order: issued here
            # too — _TEACHER_AHEAD_ALL — the host reaches the student's labeled pass 10 ms later and the main lane idles;
            # measured 5 ms worse.)  The student's passes wait for the optimizer as before.
            from ..pcdet.pfe import VoxelSetAbstraction
            for s in lanes.streams[1:]:
                s.wait_event(ahead[0])
                s.wait_event(ahead[1])
            VoxelSetAbstraction.fps_stream = lanes.stream(2)
            try:
                with torch.cuda.stream(lanes.stream(2)):
                    self._issue_geometry(lab_dict, unlab_dict)
            finally:
                VoxelSetAbstraction.fps_stream = None
            for m in self.unlab_ssl_modules:
                if hasattr(m, 'issue_early') and str(getattr(m, 'ssl_obj_attr', '')).startswith('teacher') and \
                        (_TEACHER_AHEAD_ALL or getattr(m, 'early_before_geometry', False)):
                    lanes.run(m, self, unlab_dict, method='issue_early')
            lanes.fork()
            self._share_2d_trunk(lab_dict, unlab_dict, lane_mode)
        elif branches_first:
            # Scheduling only ('branches'): what needs no 3D geometry — the student's shared 2D trunk and the
            # teacher's 2D inference, both on the 2D lane — is issued BEFORE the geometry, whose size read-backs
            # wait for the tail of the previous iteration on the main stream: the host issues these while it would
            # otherwise idle there.
            lanes.fork()
            self._share_2d_trunk(lab_dict, unlab_dict, lane_mode)
            for m in self.unlab_ssl_modules:
                if getattr(m, 'early_before_geometry', False) and hasattr(m, 'issue_early'):
                    lanes.run(m, self, unlab_dict, method='issue_early')
        if ahead is None:
            self._issue_geometry(lab_dict, unlab_dict)
        if not branches_first:
            if lanes is not None:
                lanes.fork()          # inputs, weights and the prefetched geometry live on the main stream
            self._share_2d_trunk(lab_dict, unlab_dict, lane_mode)
        unlab_modules = list(self.unlab_ssl_modules)
        if lanes is not None and lanes.mode == 'glue':
            # The teacher's inference passes read nothing but the raw unlabeled batch: issue them first, so
            # that the glue that consumes them can run (on its side stream) while the main stream works
            # through the supervised passes and their early backward.  Values do not depend on the order:
            # the teacher is in eval mode and draws no random numbers.
            hoisted = [m for m in unlab_modules if getattr(m, 'hoistable', False) and
                       str(getattr(m, 'ssl_obj_attr', '')).startswith('teacher')]
            hoisted = sorted(hoisted, key=lambda m: getattr(m, 'has_readback', False))      # read-back last
            for m in hoisted:
                unlab_dict = run(m, unlab_dict)
            unlab_modules = [m for m in unlab_modules if m not in hoisted]
        early = lanes is None and getattr(self, 'early_backward', False) and torch.is_grad_enabled()
        curr_ssl_weight = self._get_curr_ssl_weight()
        sup_per_lane = lanes is not None and lanes.mode == 'branches' and _SUP_BWD_PER_LANE and torch.is_grad_enabled() \
            and getattr(self, 'early_backward', False)
        sup_2d_done = []
        for m in self.lab_ssl_modules:
            if sup_per_lane and lanes.lane_of(m) == 1:
                lab_dict = self._sup_backward_on_lane(lanes, m, lab_dict, sup_2d_done)
            else:
                lab_dict = self._run_and_backprop(run, m, lab_dict, early, curr_ssl_weight)
        lab_done = None
        if lanes is not None:
            lab_done = torch.cuda.Event()           # the labeled forward passes of the main lane are behind this
            lab_done.record(lanes.main)
        if lanes is not None and not sup_2d_done:
            lanes.join(lab_dict['sup_losses'], lab_dict['ssl_losses'])
        if getattr(self, 'early_backward', False) and torch.is_grad_enabled():
            # Scheduling only: d(sum of losses) = sum of d(losses), so the supervised part can be
            # back-propagated now.  Its (GPU-bound) backward then runs underneath the host-bound
            # teacher / pseudo-label phase instead of after it, and its graph is freed early.
            # Gradients accumulate in the flat arena (zeroed at the start of the iteration).
            sup = self._collapse_losses(dict(lab_dict['sup_losses']))
            terms = [v for k, v in sup.items() if 'loss' in k and v.requires_grad]
            if terms:
                from ..spconv.ops import deferred_weight_grads
                with deferred_weight_grads():
                    sum(terms).backward()
                for ev in sup_2d_done:    # the hook reads every gradient that exists: the 2D lane's own backward included
                    torch.cuda.current_stream().wait_event(ev)
                hook = getattr(self, 'after_partial_backward', None)
                if hook is not None:      # e.g. FlatGradDDP.collect
                    hook()
            lab_dict['sup_losses'] = {k: v.detach() for k, v in sup.items()}
        if lanes is not None and sup_2d_done:
            lanes.join(lab_dict['sup_losses'], lab_dict['ssl_losses'])
        if getattr(self, 'issue_early', True) and _ISSUE_EARLY:
            # Scheduling only: every pass of the unlabeled chain that can start before its inputs from the chain
            # exist (teacher inference up to its read-back; the label-independent trunk of the student's pass) is
            # issued now, in chain order, so that the read-backs further down find the device done.
            for m in unlab_modules:
                if hasattr(m, 'issue_early'):
                    if lanes is not None and _TRUNK_ON_2D_LANE and lanes.mode == 'branches' and lab_done is not None and \
                            getattr(m, 'trunk_may_change_lane', False):
                        # Scheduling only: the label-independent trunk of the student's unlabeled 3D pass on the 2D lane
                        # (idle between the labeled 2D pass and the pseudo-label 2D module), beside the supervised
                        # backward on the main lane instead of behind it.  It reads the student's BatchNorm statistics
                        # after the labeled forward (the event), as in the one-lane order; its backward runs where its
                        # forward ran.
                        lanes.stream(1).wait_event(lab_done)
                        lanes.run(m, self, unlab_dict, method='issue_early', lane=1)
                    elif lanes is not None:
                        lanes.run(m, self, unlab_dict, method='issue_early')
                    else:
                        rng.enter(m)
                        m.issue_early(self, unlab_dict)
        early_2d = lanes is not None and lanes.mode == 'branches' and _EARLY_2D_BWD and torch.is_grad_enabled() and \
            getattr(self, '_deferred', None) and getattr(self, 'early_backward', False)
        last_2d = None
        if early_2d:
            from .ssl_modules import HardPseudoLabel_2D
            cands = [m for m in unlab_modules if isinstance(m, HardPseudoLabel_2D)]
            users = [m for m in unlab_modules if str(getattr(m, 'ssl_obj_attr', '')).endswith('student.detector_2d')]
            last_2d = cands[-1] if cands and users and users[-1] is cands[-1] else None
        split3d = None
        if last_2d is not None and _2D_INSIDE_3D:
            # Scheduling only: the student's last 2D module (+ its losses and the deferred 2D trunk backward, all on the
            # 2D lane) is issued between the ISSUE of its 3D neighbour and that neighbour's read-back when it does not read
            # the neighbour's outputs — the host issues it while the device works through the 3D heads, and nothing stands
            # between the read-back and the consistency glue that gates the last backward.
            from .ssl_modules import Opd_HardPseudoLabel_3D
            i = unlab_modules.index(last_2d)
            prev = unlab_modules[i - 1] if i > 0 else None
            if isinstance(prev, Opd_HardPseudoLabel_3D) and prev.out_bboxes_key is not None and \
                    not str(last_2d.target_bboxes_key).startswith(prev.target_batch_dict_key + '.' + prev.out_bboxes_key):
                split3d = prev
        for m in unlab_modules:
            if m is split3d:
                unlab_dict = lanes.run(m, self, unlab_dict, method='forward_issue')
                unlab_dict = self._early_2d_backward(lanes, last_2d, unlab_dict, curr_ssl_weight)
                unlab_dict = lanes.run(m, self, unlab_dict, method='forward_finish')
            elif m is last_2d:
                if split3d is None:
                    unlab_dict = self._early_2d_backward(lanes, m, unlab_dict, curr_ssl_weight)
            else:
                unlab_dict = self._run_and_backprop(run, m, unlab_dict, early, curr_ssl_weight)
        if lanes is not None:
            lanes.join(unlab_dict['ssl_losses'], lab_dict['ssl_losses'])
        losses = dict()
        losses.update(add_prefix(lab_dict['sup_losses'], 'sup'))
        ssl_losses = dict()
        ssl_losses.update(add_prefix(lab_dict['ssl_losses'], 'lab'))
        ssl_losses.update(add_prefix(unlab_dict['ssl_losses'], 'unlab'))
        vis_dict, log_vars_dict = dict(), dict()
        for tag, d in (('lab', lab_dict), ('unlab', unlab_dict)):
            vis_dict.update({'%s/%s' % (tag, k): v for k, v in d.get('vis', {}).items()})
            log_vars_dict.update({'%s/%s' % (tag, k): v for k, v in d.get('log_vars', {}).items()})
        ref = list(losses.values())[0] if len(losses) else list(ssl_losses.values())[0]
        losses['ssl.weight'] = torch.full((), float(curr_ssl_weight), dtype=ref.dtype, device=ref.device)
        ssl_losses = self._collapse_losses(ssl_losses)
        for k in ssl_losses.keys():
            if '.metrics' not in k and '.acc' not in k:
                ssl_losses[k] = ssl_losses[k] * curr_ssl_weight
        losses.update(add_prefix(ssl_losses, 'ssl'))
        losses['vis'] = vis_dict
        losses['log_vars'] = log_vars_dict
        losses['ssl.ema_decay'] = torch.full((), float(self._get_curr_ema_decay()), dtype=ref.dtype,
                                             device=ref.device)
        # the EMA runs INSIDE forward_train, before this iteration's backward / step
        # (ssl.py:348): teacher_t = d teacher_{t-1} + (1-d) student_{t-1, post-step}
        from ..bn_relu import flush_counters
        flush_counters()
        with torch.no_grad():
            self._update_teacher()
        if dev.type == 'cuda':
            self._ema_done = torch.cuda.Event()      # what the NEXT iteration's teacher passes wait for
            self._ema_done.record()
        self._lanes = None
        rng.finish()
        return losses

    @staticmethod
    def _collate(batches):
        out = dict()
        for k in batches[0].keys():
            out[k] = [sample for batch in batches for sample in batch[k]]
            if k in ['img']:
                out[k] = torch.stack(out[k], dim=0)
        return out

    def forward_test(self, **kwargs):
        for name in ['img', 'points', 'img_metas']:
            if name in kwargs:
                assert isinstance(kwargs[name], list)
                kwargs[name] = kwargs[name][0]
        return self.simple_test(**kwargs)

    def simple_test(self, **kwargs):
        """ssl.py:361-371"""
        teacher_res = self.teacher.simple_test(**dict(kwargs))
        student_res = self.student.simple_test(**dict(kwargs))
        return [dict(teacher=t, student=s) for t, s in zip(teacher_res, student_res)]
