"""Data-parallel gradient exchange for the DetMatch step.

One process per GPU; the ONLY data-path collective of an iteration is the average of the
student's gradients.  The reference wraps the model in MMDistributedDataParallel with
find_unused_parameters=True (mmdet3d/apis/ssl_train.py:73-80): 25 MB buckets and a graph walk
per iteration to find parameters that got no gradient.

Here the gradients of all trainable parameters live in ONE flat fp32 arena (`p.grad` are views),
zeroed at the start of each step, so a parameter that gets no gradient contributes zeros by
construction — no graph walk.  The arena is cut into a few large buckets (default 64 MiB: xGMI is
point-to-point, ring steps are per-link bound, so fewer/larger messages win) in reverse
registration order.

Two ways of getting gradients into the arena:
  mode='collect' (default)  `.grad` starts as None, autograd keeps the first gradient of a
      parameter without any kernel, and `collect()` adds all produced gradients into the arena with
      batched multi-tensor launches (a dozen launches instead of one accumulation kernel per
      parameter per backward pass: ~600 parameters x 2 passes per DetMatch iteration); `finish()`
      then all-reduces the buckets.  216 MB over xGMI is ~1 % of an iteration, so not overlapping
      it with backward costs nothing measurable.
  mode='hooks'   `.grad` are views of the arena, autograd accumulates in place, and a bucket's
      all-reduce is issued asynchronously as soon as its last gradient of the (final) backward pass
      has been produced, overlapping the rest of backward.  Buckets are ALWAYS issued in index
      order (bucket b only after 0..b-1): which parameters get a gradient depends on the data
      (pseudo-label / match counts), so a "ready first, sent first" order would differ between
      ranks and pair bucket i of one rank with bucket j of another.

Liveness (the reference's find_unused_parameters=True + mmcv zero_grad semantics,
mmdet3d/apis/ssl_train.py:65-69): a parameter that NO rank has ever produced a gradient for keeps
`.grad is None` in the reference, so torch optimizers skip it (no weight decay, no state).  The
arena ends in one float flag per parameter ("got a gradient this step on this rank"); it rides in
the last bucket's all-reduce, so `ever` (sticky, device-resident) is the same on all ranks without
an extra collective, and the fused optimizer kernels take it as a per-16-byte-block mask
(`live_mask`).  Once every flag is set (probed through a pinned, event-guarded copy: no host
sync) the mask is dropped.

Exchange: `all_reduce` per bucket (default), or `exchange='rs_ag'` (env DM_GRAD_EXCHANGE=rs_ag):
reduce_scatter_tensor + all_gather_into_tensor per bucket, the direct algorithm over the 7 xGMI
links SURVEY §8(e) names (needs the nccl backend; buckets are padded to a multiple of the world
size).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


def _host_staged(group=None):
    """True when the process group cannot take device tensors (gloo on this image): collectives on device tensors
    are then staged through host copies — the TEST path that runs several ranks on one GPU (bench.py:
    DM_FORCE_DEVICE + DM_DIST_BACKEND=gloo, tests/test_multirank_gpu.py); RCCL takes the device tensors as they are."""
    return dist.is_initialized() and dist.get_backend(group) == 'gloo'


class _Done(object):
    def wait(self):
        return True


def all_reduce(t, group=None, async_op=False, op=None):
    """dist.all_reduce for device tensors under either backend (see _host_staged)."""
    kw = {} if op is None else dict(op=op)
    if t.is_cuda and _host_staged(group):
        h = t.detach().cpu()
        dist.all_reduce(h, group=group, **kw)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, group=group, async_op=async_op, **kw)


def all_gather_object(obj, group=None):
    """[obj of rank 0, obj of rank 1, ...] on every rank (torch.distributed.all_gather_object; the validation results)."""
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def broadcast(t, src, group=None):
    if t.is_cuda and _host_staged(group):
        h = t.detach().cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src, group=group)


class _Chain2(object):
    """Work handle of two dependent steps (reduce-scatter, then all-gather) done eagerly."""

    def wait(self):
        return True


def reduce_scatter_then_all_gather(buf, shard, group=None):
    """buf <- all-gather(reduce-scatter(buf)): the `rs_ag` exchange of one bucket.  RCCL: the two collectives,
    asynchronous, on the process group's stream (the all-gather is ordered behind the reduce-scatter there).
    gloo (CPU tests, and several ranks on one GPU): gloo has no reduce_scatter_tensor — every rank reduces the whole
    bucket, keeps ITS shard (what a reduce-scatter leaves it with) and the shards are gathered back, so that the
    shard arithmetic (bucket padding to a multiple of the world size, shard offsets, rank order) is exercised."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) != 'gloo':
        dist.reduce_scatter_tensor(shard, buf, group=group, async_op=True)
        return dist.all_gather_into_tensor(buf, shard, group=group, async_op=True)
    rank = dist.get_rank(group)
    n = shard.numel()
    assert buf.numel() == n * world, 'bucket not padded to a multiple of the world size'
    host = buf.detach().cpu() if buf.is_cuda else buf.detach().clone()
    dist.all_reduce(host, group=group)
    mine = host[rank * n:(rank + 1) * n].clone()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    shard.copy_(mine)
    buf.copy_(torch.cat(parts))
    return _Chain2()


class FlatGradDDP(nn.Module):

    def __init__(self, module, params=None, bucket_bytes=64 << 20, process_group=None,
                 broadcast=True, mode='collect', exchange=None, always_exchange=False):
        super().__init__()
        # issue the collectives even in a one-rank group (they are identities there): lets a single GPU
        # execute the exact exchange code path of an N-rank job (tests/test_ssl_gpu.py)
        self.always_exchange = always_exchange
        assert mode in ('collect', 'hooks')
        self.mode = mode
        self.exchange = exchange or os.environ.get('DM_GRAD_EXCHANGE', 'all_reduce')
        assert self.exchange in ('all_reduce', 'rs_ag')
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in (params if params is not None else module.parameters())
                       if p.requires_grad]
        # True when the arena holds every trainable parameter: its norm IS the global norm
        self.covers_all_clipped = params is None
        assert self.params, 'nothing to train'
        dev = self.params[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in self.params)
        # reverse order ~ the order in which backward produces gradients; every tensor starts on a
        # 4-element (16-byte) boundary so that the parameter arena, the optimizer state and this
        # arena can be walked index-aligned by float4 kernels
        order = list(range(len(self.params)))[::-1]
        pad = lambda n: (n + 3) // 4 * 4
        n_grad = sum(pad(p.numel()) for p in self.params)
        # tail: one "got a gradient" flag per parameter, padded so that rs_ag can cut every bucket
        # into world equal shards
        n_all = n_grad + pad(len(self.params))
        n_all = (n_all + 4 * self.world - 1) // (4 * self.world) * (4 * self.world)
        self.flat_all = torch.zeros(n_all, dtype=torch.float32, device=dev)
        self.flat = self.flat_all[:n_grad]               # the gradients (what clip / optimizers see)
        self.used = self.flat_all[n_grad:n_grad + len(self.params)]
        self.ever = torch.zeros(len(self.params), dtype=torch.bool, device=dev)   # sticky, all ranks
        self.order = [self.params[i] for i in order]     # arena order
        self.index = {id(p): i for i, p in enumerate(self.order)}
        self._blocks = torch.tensor([pad(p.numel()) // 4 for p in self.order], dtype=torch.int64,
                                    device=dev)
        self.n_blocks = n_grad // 4
        self.block_live = None      # uint8 per 4-element block while some parameter is still dead
        self._all_live = False
        self._probe = None
        self._mark_cache = {}
        self._fired = set()
        self._local_ever = set()    # arena indices that got a gradient on THIS rank (host knowledge)
        self._dead_cache = None
        self.offset = {}                                 # id(param) -> first element
        self.flat_params = None                          # set by build_param_arena / SSL.build_arenas
        self.buckets = []          # (start, end) element ranges of self.flat
        self._bucket_of = {}
        self._pending = []
        off, b_start, cap = 0, 0, max(1, bucket_bytes // 4)
        self._view = {}
        for p in self.order:
            self._view[id(p)] = self.flat[off:off + p.numel()].view_as(p)
            p.grad = self._view[id(p)] if mode == 'hooks' else None
            self.offset[id(p)] = off
            self._bucket_of[id(p)] = len(self.buckets)
            off += pad(p.numel())
            if off - b_start >= cap and off % (4 * self.world) == 0:
                self.buckets.append((b_start, off))
                b_start = off
        self.buckets.append((b_start, n_all))            # last bucket: rest + the flag tail
        self._need = [0] * len(self.buckets)
        for p in self.params:
            self._need[self._bucket_of[id(p)]] += 1
        self._left = list(self._need)
        self._sent = [False] * len(self.buckets)
        self._next = 0
        self._armed = False
        self._shards = None
        if mode == 'hooks':
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)
        if broadcast and self.world > 1:
            self.broadcast_parameters()

    @torch.no_grad()
    def build_param_arena(self):
        """Re-home the trainable parameters in a flat arena laid out exactly like the gradient
        arena (fused optimizer kernels walk both index-aligned)."""
        flat = torch.zeros_like(self.flat)
        for p in self.order:
            off = self.offset[id(p)]
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.flat_params = flat
        return flat

    def check_param_arena(self):
        base = self.flat_params.data_ptr()
        for p in self.order:
            if p.data_ptr() != base + 4 * self.offset[id(p)]:
                return False
        return True

    # ---- state sync ------------------------------------------------------------------
    @torch.no_grad()
    def broadcast_parameters(self, src=0):
        """Rank `src`'s parameters AND buffers to everyone, two flat messages."""
        for tensors in ([p.data for p in self.module.parameters()],
                        [b.data for b in self.module.buffers()]):
            for dtype in (torch.float32, torch.int64):
                ts = [t for t in tensors if t.dtype == dtype]
                if not ts:
                    continue
                flat = torch.cat([t.reshape(-1) for t in ts])
                broadcast(flat, src, group=self.group)
                off = 0
                for t in ts:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    # ---- step protocol: zero_grad() -> backward -> finish() -------------------------------
    def zero_grad(self, arm=True):
        """arm=False (hooks mode): zero only; the bucket hooks stay quiet until arm() (several
        backward passes accumulate into the arena, only the last one may trigger the exchange)."""
        self.flat_all.zero_()
        self._fired = set()
        self._next = 0
        if self.mode == 'collect':
            for p in self.params:
                p.grad = None
        else:
            for p in self.params:      # optimizers / user code may have replaced .grad
                if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or \
                        p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * 4:
                    raise RuntimeError('a gradient left the flat arena (zero_grad(set_to_none=True)?)')
        self._left = list(self._need)
        self._sent = [False] * len(self.buckets)
        self._pending = []
        self._armed = arm and self.mode == 'hooks'

    def collect(self):
        """collect mode: add the gradients autograd has produced so far into the arena (batched
        multi-tensor adds) and release them.  Call after every backward pass."""
        from .. import _lib
        _lib.wait_pending_grads()          # weight gradients issued on the side stream (chain.SIDE_WGRAD)
        if self.mode != 'collect':
            return
        ps = [p for p in self.order if p.grad is not None and p.grad.data_ptr() != self._view[id(p)].data_ptr()]
        if ps:
            self._mark([self.index[id(p)] for p in ps])
            torch._foreach_add_([self._view[id(p)] for p in ps], [p.grad for p in ps])
            if self.flat.is_cuda:      # gradients may have been produced on another stream (2D lane)
                cur = torch.cuda.current_stream(self.flat.device)
                for p in ps:
                    p.grad.record_stream(cur)
        for p in ps:
            p.grad = None

    def _mark(self, idxs):
        """used[idxs] = 1 (index tensors cached per distinct set: no H2D copy in the steady state)."""
        if not idxs:
            return
        if not self._local_ever.issuperset(idxs):
            self._local_ever.update(idxs)
            self._dead_cache = None
        key = tuple(idxs)
        t = self._mark_cache.get(key)
        if t is None:
            if len(self._mark_cache) > 64:
                self._mark_cache.clear()
            t = self._mark_cache[key] = torch.tensor(idxs, dtype=torch.int64, device=self.flat.device)
        self.used.index_fill_(0, t, 1.0)

    def arm(self):
        if self.mode == 'hooks':
            self._left = list(self._need)
            self._armed = True

    def _launch(self, b):
        s, e = self.buckets[b]
        self._sent[b] = True
        if self.world <= 1 and not self.always_exchange:
            return
        self.n_collectives = getattr(self, 'n_collectives', 0) + (2 if self.exchange == 'rs_ag' else 1)
        buf = self.flat_all[s:e]
        if self.exchange == 'rs_ag':
            if self._shards is None:
                self._shards = [torch.empty((e2 - s2) // self.world, dtype=torch.float32,
                                            device=self.flat.device) for s2, e2 in self.buckets]
            # same stream of the process group: the all-gather runs after the reduce-scatter
            self._pending.append(reduce_scatter_then_all_gather(buf, self._shards[b], self.group))
        else:
            self._pending.append(all_reduce(buf, group=self.group, async_op=True))

    def _launch_ready(self, limit):
        """Issue buckets strictly in index order while the next one is complete."""
        while self._next < limit and self._left[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _on_grad(self, p):
        self._fired.add(self.index[id(p)])
        if not self._armed:
            return
        self._left[self._bucket_of[id(p)]] -= 1
        # the last bucket carries the liveness flags, written in finish(): never from a hook
        self._launch_ready(len(self.buckets) - 1)

    def finish(self):
        """Issue the remaining buckets, wait for all, average; afterwards `p.grad` are the arena
        views (what clip / optimizers that are not fused read)."""
        self._armed = False
        if self.mode == 'collect':
            self.collect()
            for p in self.order:
                p.grad = self._view[id(p)]
        else:
            self._mark(sorted(self._fired))
        for b in range(self._next, len(self.buckets)):       # the rest, in index order
            self._launch(b)
        self._next = len(self.buckets)
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.world > 1:
            self.flat.div_(self.world)
            self._dead_cache = None          # other ranks' flags arrived with the last bucket
        self._update_live()

    # ---- liveness ---------------------------------------------------------------------
    def _update_live(self):
        if self._all_live:
            return
        if self._probe is not None:          # did an earlier step already see every flag set?
            host, ev = self._probe
            if ev is None or ev.query():
                if bool(host.item()):
                    self._all_live, self.block_live, self._probe = True, None, None
                    return
                self._probe = None
        torch.logical_or(self.ever, self.used > 0, out=self.ever)
        self.block_live = torch.repeat_interleave(self.ever.to(torch.uint8), self._blocks,
                                                  output_size=self.n_blocks)
        if self._probe is None:
            allv = self.ever.all().reshape(1)
            if self.flat.is_cuda:
                host = torch.empty(1, dtype=torch.bool, pin_memory=True)
                host.copy_(allv, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._probe = (host, ev)
            else:
                self._probe = (allv.clone(), None)

    def live_mask(self, lo, hi):
        """uint8 per 4-element block of arena range [lo, hi) (0 = the parameter never received a
        gradient on any rank), or None when every parameter is live."""
        if self.block_live is None:
            return None
        return self.block_live[lo // 4:(hi + 3) // 4]

    def dead_params(self):
        """Parameters no rank has produced a gradient for so far (used by non-fused optimizers and
        state_dict only).  One rank: exact from host knowledge (which parameters fired), no read-back.
        Several ranks: a parameter may be live on another rank only, so the all-reduced flags are read back —
        once per exchange (cached until the next finish()), and only while some parameter is still dead."""
        if self._all_live:
            return []
        if self.world <= 1:
            if self._dead_cache is None:
                self._dead_cache = [p for i, p in enumerate(self.order) if i not in self._local_ever]
            return self._dead_cache
        if self._dead_cache is None:
            ever = self.ever.cpu().tolist()
            self._dead_cache = [p for p, e in zip(self.order, ever) if not e]
        return self._dead_cache

    def clip_coef(self, max_norm, norm_type=2):
        """-> (total_norm, coef) device scalars of clip_grad_norm_; nothing is scaled (the fused
        optimizer kernels apply coef while they read the gradients)."""
        assert norm_type == 2
        total = torch.linalg.vector_norm(self.flat)
        return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)

    def clip_grad_norm_(self, max_norm, norm_type=2):
        """Global clip over the arena: one norm kernel + one scale kernel."""
        total, coef = self.clip_coef(max_norm, norm_type)
        self.flat.mul_(coef)
        return total

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def train_step(self, *args, **kwargs):
        return self.module.train_step(*args, **kwargs)
