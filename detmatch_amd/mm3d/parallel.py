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
      has been produced, overlapping the rest of backward.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class FlatGradDDP(nn.Module):

    def __init__(self, module, params=None, bucket_bytes=64 << 20, process_group=None,
                 broadcast=True, mode='collect'):
        super().__init__()
        assert mode in ('collect', 'hooks')
        self.mode = mode
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
        self.flat = torch.zeros(sum(pad(p.numel()) for p in self.params), dtype=torch.float32,
                                device=dev)
        self.order = [self.params[i] for i in order]     # arena order
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
            if off - b_start >= cap:
                self.buckets.append((b_start, off))
                b_start = off
        if off > b_start:
            self.buckets.append((b_start, off))
        self._need = [0] * len(self.buckets)
        for p in self.params:
            self._need[self._bucket_of[id(p)]] += 1
        self._left = list(self._need)
        self._sent = [False] * len(self.buckets)
        self._armed = False
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
                dist.broadcast(flat, src, group=self.group)
                off = 0
                for t in ts:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    # ---- step protocol: zero_grad() -> backward -> finish() -------------------------------
    def zero_grad(self, arm=True):
        """arm=False (hooks mode): zero only; the bucket hooks stay quiet until arm() (several
        backward passes accumulate into the arena, only the last one may trigger the exchange)."""
        self.flat.zero_()
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
        if self.mode != 'collect':
            return
        ps = [p for p in self.order if p.grad is not None and p.grad.data_ptr() != self._view[id(p)].data_ptr()]
        if ps:
            torch._foreach_add_([self._view[id(p)] for p in ps], [p.grad for p in ps])
            if self.flat.is_cuda:      # gradients may have been produced on another stream (2D lane)
                cur = torch.cuda.current_stream(self.flat.device)
                for p in ps:
                    p.grad.record_stream(cur)
        for p in ps:
            p.grad = None

    def arm(self):
        if self.mode == 'hooks':
            self._left = list(self._need)
            self._armed = True

    def _launch(self, b):
        s, e = self.buckets[b]
        self._sent[b] = True
        if self.world > 1:
            self._pending.append(dist.all_reduce(self.flat[s:e], group=self.group, async_op=True))

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self._bucket_of[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0 and not self._sent[b]:
            self._launch(b)

    def finish(self):
        """Issue the remaining buckets, wait for all, average; afterwards `p.grad` are the arena
        views (what clip / optimizers that are not fused read)."""
        self._armed = False
        if self.mode == 'collect':
            self.collect()
            for p in self.order:
                p.grad = self._view[id(p)]
        for b in range(len(self.buckets)):
            if not self._sent[b]:
                self._launch(b)
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.world > 1:
            self.flat.div_(self.world)

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
