"""hipGraph replay of shape-static, sync-free sections: StaticSection (inference, no autograd) and TrainSection
(forward graph + backward graph behind one autograd node).

A section is a function of a few tensors whose launches depend on the shapes only: the teacher's 2D trunk
(ResNet + FPN + RPN convolutions, ~150 launches) or its BEV backbone + dense-head convolutions.  The host
issues such a section in 2-4 ms of Python although the device needs about the same time, and the step is
host-bound right behind it (the 3D passes with their data-dependent sizes cannot be captured).  Captured
once per input signature, the section costs one copy of the inputs into static buffers and one
hipGraphLaunch; every kernel inside is the very same C-ABI launch on the capture stream.  The packed copies of
the weights are NOT in the graph (round 4): its convolutions read the replay stream's cached copies, which
`dense_conv.ensure_fresh` brings up to date with ONE launch in front of a replay (the arenas never move, the
cached copies never move, the section keeps them alive); only weights with per-call scales (BatchNorm folds of
a moving teacher) keep recorded packs.

Rules a section must keep (checked by the capture itself, which fails on a violation): no device->host
read, no allocation outside torch's allocator, no launch on another stream.  The outputs are static
tensors, overwritten by the next replay of the same signature: consume them before calling again (the
callers use them within the iteration, on the same stream).

Measured on the DetMatch iteration (same box, alternated, 40 steps each): the teacher's 2D trunk as one
graph takes its host time from 3.1 to 1.4 ms, but the phase is device-bound (5.0 ms of kernels, and the
recorded per-weight packs and BatchNorm folds add 0.6 ms to them): 94.5 / 95.8 / 94.3 ms with the graph against
93.8 / 94.9 / 93.4 ms without.  The step's idle time is not in its static sections, so the capture is
OPT-IN: `DM_HIPGRAPH=1`; by default every section is a plain call.
"""
import os
import time

import torch

TIMING = os.environ.get('DM_GRAPH_TIMING', '0') == '1'

ENABLED = os.environ.get('DM_HIPGRAPH', '0') == '1'
_WARMUP_CALLS = 2          # eager calls of a signature before it is captured (allocator / pack caches settle)


def _flatten(out, acc):
    if isinstance(out, torch.Tensor):
        acc.append(out)
    elif isinstance(out, (list, tuple)):
        for o in out:
            _flatten(o, acc)
    elif isinstance(out, dict):
        for o in out.values():
            _flatten(o, acc)
    return acc


class StaticSection(object):
    """section = StaticSection(fn); out = section(*tensors).  `fn` must be a pure function of the tensors'
    VALUES and of module parameters that live at fixed addresses."""

    def __init__(self, fn, name='section'):
        self.fn = fn
        self.name = name
        self.entries = {}
        self.replays = 0
        self.captures = 0

    def __call__(self, *tensors, frozen=False):
        """frozen=True: the caller vouches that nothing reachable from `fn` requires grad (a teacher whose
        parameters are all requires_grad=False runs with autograd enabled but records nothing)."""
        if not ENABLED or (torch.is_grad_enabled() and not frozen) or not all(t.is_cuda for t in tensors) \
                or any(t.requires_grad for t in tensors):
            return self.fn(*tensors)
        with torch.no_grad():
            return self._run(*tensors)

    def _run(self, *tensors):
        key = tuple((tuple(t.shape), t.dtype, tuple(t.stride()), t.device.index) for t in tensors)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = dict(calls=0, graph=None)
        if e['graph'] is None:
            e['calls'] += 1
            if e['calls'] <= _WARMUP_CALLS:
                return self.fn(*tensors)
            self._capture(e, tensors)
        for s, t in zip(e['static_in'], tensors):
            s.copy_(t)
        from . import dense_conv
        dense_conv.ensure_fresh(tensors[0].device, e['keep'])      # the graph's convolutions read the stream's cached packs
        e['graph'].replay()
        self.replays += 1
        return e['out']

    def _capture(self, e, tensors):
        static_in = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device) for t in tensors]
        for s, t in zip(static_in, tensors):
            s.copy_(t)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=tensors[0].device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):       # once more on a side stream: whatever is keyed by the stream settles
            self.fn(*static_in)
        cur.wait_stream(side)
        from . import dense_conv, _lib
        g = torch.cuda.CUDAGraph()
        dense_conv.ensure_fresh(tensors[0].device)
        dense_conv.CAPTURING[0] = True      # constant-scale weight packs come from the replay stream's cache; folded-BatchNorm maps are recorded
        dense_conv.CAPTURE_STREAM[0] = _lib.raw_stream()
        del dense_conv.CAPTURE_KEEP[:]
        try:
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                out = self.fn(*static_in)
        finally:
            dense_conv.CAPTURING[0] = False
            dense_conv.CAPTURE_STREAM[0] = None
        e['keep'] = list(dense_conv.CAPTURE_KEEP)
        del dense_conv.CAPTURE_KEEP[:]
        e['graph'], e['static_in'], e['out'] = g, static_in, out
        self.captures += 1


# ---- sections that record autograd ---------------------------------------------------------------------------
# The student's 2D trunk (ResNet-50 + FPN + RPN convolutions over the iteration's four images) and each student
# pass's BEV backbone + anchor-head convolutions are shape-static chains of 150 / 60 launches forward and about
# three times as many backward; Python issues each launch in 15-25 us although the device needs 2-5 us for the
# small ones, and the iteration is paced by the host.  A TrainSection captures such a chain once per
# input signature as TWO graphs — forward, and backward (torch.autograd.grad over the recorded autograd graph,
# issued by autograd's device thread into the capturing stream) — and puts ONE autograd node in their place:
# forward = copy the inputs into the static buffers + replay, backward = copy the incoming gradients + replay +
# hand the static parameter gradients to autograd.  Every kernel inside is the same C-ABI launch as in the plain
# call (weight packs included: a replay reads the current weights), so values are those of the plain call.
#
# A section may be called several times per iteration (labeled / unlabeled student pass) before any backward has
# run: every call in flight owns its own INSTANCE (static buffers + the two graphs).  An instance is free again
# once its backward has replayed, or at new_iteration().
TRAIN_ENABLED = os.environ.get('DM_HIPGRAPH_TRAIN', os.environ.get('DM_HIPGRAPH', '0')) == '1'
_MAX_INSTANCES = 4
_ALL_TRAIN_SECTIONS = []


def new_iteration():
    """Every instance of every TrainSection is free again (called where an iteration starts: no forward of the
    previous iteration still waits for its backward)."""
    for ref in _ALL_TRAIN_SECTIONS:
        sec = ref()
        if sec is not None:
            for insts in sec.instances.values():
                for inst in insts:
                    inst.busy = False


class _Instance(object):
    __slots__ = ('fwd', 'bwd', 'static_in', 'outs', 'diff', 'gouts', 'gins', 'busy', 'fn_cls', 'keep')


class TrainSection(object):
    """section = TrainSection(fn, modules); outs = section(*tensors) -> tuple of tensors.

    fn(*tensors) -> tuple of tensors must be a pure function of the tensors' VALUES and of the parameters and
    buffers of `modules` (read through the modules' attributes at call time; BatchNorm running statistics may be
    updated in place), free of host reads and of launches on other streams.  Inputs keep their requires_grad."""

    def __init__(self, fn, modules, name='section'):
        import weakref
        self.fn = fn
        self.modules = [m for top in modules for m in top.modules()]
        self.slots = [(m, k) for m in self.modules for k, p in m._parameters.items()
                      if p is not None and p.requires_grad]
        self.params = [m._parameters[k] for m, k in self.slots]
        self.buffers = [b for m in self.modules for b in m._buffers.values() if b is not None]
        self.name = name
        self.instances = {}
        self.calls = {}
        self.captures = 0
        self.replays = 0
        self.bwd_replays = 0
        self.fallbacks = 0
        self.t_replay = self.t_replay_bwd = 0.0      # host seconds inside hipGraphLaunch (DM_GRAPH_TIMING=1)
        _ALL_TRAIN_SECTIONS.append(weakref.ref(self))

    def __call__(self, *tensors):
        if not TRAIN_ENABLED or not torch.is_grad_enabled() or not all(t.is_cuda for t in tensors):
            return self.fn(*tensors)
        key = tuple((tuple(t.shape), t.dtype, tuple(t.stride()), t.device.index, t.requires_grad) for t in tensors)
        n = self.calls[key] = self.calls.get(key, 0) + 1
        insts = self.instances.setdefault(key, [])
        inst = next((i for i in insts if not i.busy), None)
        if inst is None:
            if n <= _WARMUP_CALLS * max(1, len(insts) + 1) or len(insts) >= _MAX_INSTANCES:
                self.fallbacks += 1
                return self.fn(*tensors)
            inst = self._capture(tensors)
            insts.append(inst)
        inst.busy = True
        self.replays += 1
        return inst.fn_cls.apply(*(tuple(tensors) + tuple(self.params)))

    def _capture(self, tensors):
        from . import bn_relu, dense_conv
        if any(m._parameters[k] is not p for (m, k), p in zip(self.slots, self.params)):
            raise RuntimeError('TrainSection %s: a parameter was replaced after the section was built' % self.name)
        inst = _Instance()
        inst.busy = False
        static_in = []
        for t in tensors:
            s = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
            s.copy_(t.detach())
            static_in.append(s.requires_grad_(t.requires_grad))
        saved = [b.clone() for b in self.buffers]      # the capture-time calls must not advance the state
        # The graphs are recorded on ALIASES of the parameters (fresh leaves on the same storage).  The real
        # parameters' AccumulateGrad nodes remember the stream they were created on (the caller's), and the
        # engine would make that stream wait for the capturing one — pulling it into a capture it never leaves
        # (hipStreamEndCapture then faults instead of reporting unjoined work).
        aliases = [torch.nn.Parameter(p.detach(), requires_grad=True) for p in self.params]
        for (m, k), a in zip(self.slots, aliases):
            m._parameters[k] = a
        from . import _lib
        dense_conv.ensure_fresh(tensors[0].device)
        dense_conv.CAPTURING[0] = True        # constant-scale weight packs come from the replay stream's cache (see dense_conv)
        dense_conv.CAPTURE_STREAM[0] = _lib.raw_stream()
        dense_conv.ALIAS_OF.update({id(a): p for a, p in zip(aliases, self.params)})
        del dense_conv.CAPTURE_KEEP[:]
        bn_relu.CAPTURING[0] = True           # call counters move inside the graph
        torch.cuda.synchronize()
        try:
            side = torch.cuda.Stream(device=tensors[0].device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # one plain forward + backward: lazy initialisation stays out of the graphs
                outs = _flat_tuple(self.fn(*static_in))
                diff = [o for o in outs if o.requires_grad]
                if diff:
                    torch.autograd.grad(diff, [t for t in static_in if t.requires_grad] + aliases,
                                        [torch.zeros_like(o) for o in diff], allow_unused=True)
                del outs, diff
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for b, v in zip(self.buffers, saved):
                b.copy_(v)
            pool = torch.cuda.graph_pool_handle()
            inst.fwd, inst.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            mode = os.environ.get('DM_GRAPH_CAPTURE_MODE', 'thread_local')
            with torch.cuda.graph(inst.fwd, pool=pool, capture_error_mode=mode):
                outs = _flat_tuple(self.fn(*static_in))
            diff = [o for o in outs if o.requires_grad]
            gouts = [torch.zeros_like(o) for o in diff]
            wrt = [t for t in static_in if t.requires_grad] + aliases
            gins = ()
            if diff and wrt:
                # (autograd's device thread issues the launches into the capturing stream)
                with torch.cuda.graph(inst.bwd, pool=pool, capture_error_mode=mode):
                    gins = torch.autograd.grad(diff, wrt, gouts, allow_unused=True)
            else:
                inst.bwd = None
        finally:
            dense_conv.CAPTURING[0] = False
            dense_conv.CAPTURE_STREAM[0] = None
            for a in aliases:
                dense_conv.ALIAS_OF.pop(id(a), None)
            bn_relu.CAPTURING[0] = False
            for (m, k), p in zip(self.slots, self.params):
                m._parameters[k] = p
        torch.cuda.synchronize()
        for b, v in zip(self.buffers, saved):      # a capture executes nothing, but be explicit
            b.copy_(v)
        inst.static_in, inst.outs, inst.diff, inst.gouts = static_in, outs, diff, gouts
        inst.keep = (aliases, list(dense_conv.CAPTURE_KEEP))
        del dense_conv.CAPTURE_KEEP[:]
        grad_of = {id(w): g for w, g in zip(wrt, gins)}
        inst.gins = [grad_of.get(id(t)) if t.requires_grad else None for t in static_in] + \
                    [grad_of.get(id(a)) for a in aliases]
        inst.fn_cls = self._node(inst)
        self.captures += 1
        return inst

    def _node(self, inst):
        section = self
        n_in = len(inst.static_in)
        diff_ids = {id(o) for o in inst.diff}
        is_diff = [id(o) in diff_ids for o in inst.outs]

        class Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *args):
                for s, t in zip(inst.static_in, args[:n_in]):
                    if s.data_ptr() != t.data_ptr():
                        s.detach().copy_(t)
                from . import dense_conv
                dense_conv.ensure_fresh(inst.static_in[0].device, inst.keep[1])      # the graphs read the stream's cached packs
                t0 = time.perf_counter() if TIMING else 0.0
                inst.fwd.replay()
                if TIMING:
                    section.t_replay += time.perf_counter() - t0
                res = tuple(o.detach() for o in inst.outs)
                ctx.mark_non_differentiable(*[r for r, d in zip(res, is_diff) if not d])
                ctx.set_materialize_grads(False)
                return res

            @staticmethod
            @torch.autograd.function.once_differentiable
            def backward(ctx, *grads):
                k = 0
                for g, d in zip(grads, is_diff):
                    if not d:
                        continue
                    dst = inst.gouts[k]
                    k += 1
                    if g is None:
                        dst.zero_()
                    elif g.data_ptr() != dst.data_ptr():
                        dst.copy_(g)
                if inst.bwd is not None:
                    from . import dense_conv
                    dense_conv.ensure_fresh(inst.static_in[0].device, inst.keep[1])
                    t0 = time.perf_counter() if TIMING else 0.0
                    inst.bwd.replay()
                    if TIMING:
                        section.t_replay_bwd += time.perf_counter() - t0
                inst.busy = False
                section.bwd_replays += 1
                # the stored tensor objects themselves, not fresh views: AccumulateGrad keeps a gradient without a
                # copy only when nobody else holds it, and these buffers are rewritten by the next replay
                return tuple(inst.gins)

        return Replay


def _flat_tuple(out):
    if isinstance(out, torch.Tensor):
        return (out,)
    return tuple(_flatten(out, []))
