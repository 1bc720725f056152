"""mmdet3d/ops/spconv/ops.py — same function names / argument meaning; compute in
libdetmatch_hip.so (rulebook.hip, spconv.hip) through the C-ABI."""
import os

import torch

from .. import _lib


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops.py:20-31"""
    output_size = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) \
            // stride[i] + 1
        output_size.append(1 if kernel_size[i] == -1 else size)
    return output_size


class Rulebook(object):
    """Native rulebook of one indice_key: gather tables + reference-format pair lists."""
    __slots__ = ('outids', 'indice_pairs', 'indice_num', 'nbr_out', 'nbr_in', 'subm', 'kvol',
                 'n_in', 'n_out', 'out_shape')

    def as_reference_tuple(self):
        return self.outids, self.indice_pairs, self.indice_num


def _norm(v, ndim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * ndim


def build_rulebook(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                   subm=False, with_pairs=True):
    """Native rulebook build.  indices (N,4) int32 [b,z,y,x] on the GPU."""
    return drive_steps(build_rulebook_steps(indices, batch_size, spatial_shape, ksize, stride, padding,
                                            dilation, subm, with_pairs))


def drive_steps(gen):
    """Run a `*_steps` generator on its own: every value it asks for is read back right away."""
    try:
        ask = next(gen)
        while True:
            v = ask.reshape(-1).tolist()
            ask = gen.send(int(v[0]) if len(v) == 1 else [int(x) for x in v])
    except StopIteration as stop:
        return stop.value


def drive_steps_together(gens):
    """Run several `*_steps` generators in lockstep: the device scalars they ask for in the same round
    (voxel counts, N_out of a strided rulebook) come back in ONE device->host copy — three passes of an
    iteration cost 5 read-backs at the step boundary instead of 15.  -> list of their return values."""
    results = [None] * len(gens)
    asks = {}
    for i, g in enumerate(gens):
        try:
            asks[i] = next(g)
        except StopIteration as stop:
            results[i] = stop.value
    while asks:
        order = sorted(asks)
        flat = [asks[i].reshape(-1) for i in order]
        vals = torch.cat(flat).tolist()            # (a generator may ask for several scalars at once: one tensor)
        nxt, pos = {}, 0
        for i, f in zip(order, flat):
            v = [int(x) for x in vals[pos:pos + f.numel()]]
            pos += f.numel()
            try:
                nxt[i] = gens[i].send(v[0] if len(v) == 1 else v)
            except StopIteration as stop:
                results[i] = stop.value
        asks = nxt
    return results


def build_rulebook_steps(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                         subm=False, with_pairs=True, ws_tag='rulebook'):
    """build_rulebook as a generator: yields the device scalar whose value it needs next (N_out of a
    strided layer) and is sent that value; returns the Rulebook.  `ws_tag` names the workspace — the
    count and fill kernels of a strided layer share hash tables in it, so builds that are interleaved
    (drive_steps_together) must not share one."""
    ndim = indices.shape[1] - 1
    if ndim != 3:
        raise NotImplementedError('only 3-D sparse convolution is on the DetMatch path')
    ksize, stride, padding, dilation = (_norm(v, 3) for v in (ksize, stride, padding, dilation))
    if any(d != 1 for d in dilation):
        raise NotImplementedError('dilation != 1 is not used by VoxelBackBone8x')
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    _lib.require_device(indices)
    L = _lib.lib()
    dev = indices.device
    n = indices.shape[0]
    kvol = ksize[0] * ksize[1] * ksize[2]
    rb = Rulebook()
    rb.subm, rb.kvol, rb.n_in = bool(subm), kvol, n
    wsb = L.dm_rulebook_workspace_bytes(n, kvol)
    ws = _lib.workspace(wsb, dev, ws_tag)
    rb.indice_num = torch.empty((kvol,), dtype=torch.int32, device=dev)
    rb.indice_pairs = (torch.empty((kvol, 2, n), dtype=torch.int32, device=dev)
                       if with_pairs else None)
    spatial = [int(s) for s in spatial_shape]
    if subm:
        rb.out_shape = spatial
        rb.nbr_out = torch.empty((kvol, n), dtype=torch.int32, device=dev)
        rb.nbr_in = None
        rc = L.dm_rulebook_subm(_lib.ptr(indices), n, int(batch_size), _lib.ints(spatial),
                                _lib.ints(ksize), _lib.ptr(rb.nbr_out), _lib.ptr(rb.indice_pairs),
                                _lib.ptr(rb.indice_num), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, 'dm_rulebook_subm')
        rb.outids = indices
        rb.n_out = n
        return rb
    out_shape = get_conv_output_size(spatial, ksize, stride, padding, dilation)
    rb.out_shape = out_shape
    n_out_dev = torch.empty((1,), dtype=torch.int32, device=dev)
    args = (_lib.ptr(indices), n, int(batch_size), _lib.ints(spatial), _lib.ints(out_shape),
            _lib.ints(ksize), _lib.ints(stride), _lib.ints(padding))
    rc = L.dm_rulebook_conv_count(*args, _lib.ptr(n_out_dev), _lib.ptr(ws), ws.numel(),
                                  _lib.stream())
    _lib.check(rc, 'dm_rulebook_conv_count')
    # the one data-dependent size of the layer (reference: outInds.slice(0, 0, numActOut),
    # spconv_ops.h:139, which also costs a device->host read)
    n_out = yield n_out_dev
    rb.n_out = n_out
    rb.outids = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
    rb.nbr_out = torch.empty((kvol, n_out), dtype=torch.int32, device=dev)
    rb.nbr_in = torch.empty((kvol, n), dtype=torch.int32, device=dev)
    rc = L.dm_rulebook_conv_fill(*args, n_out, _lib.ptr(rb.outids), _lib.ptr(rb.nbr_out),
                                 _lib.ptr(rb.nbr_in), _lib.ptr(rb.indice_pairs),
                                 _lib.ptr(rb.indice_num), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, 'dm_rulebook_conv_fill')
    return rb


class CapRulebook(object):
    """A rulebook issued over the CAPACITY of its input (dm_rulebook_subm_cap / dm_rulebook_conv_cap): every count is
    still on the device.  `finish(n_in, n_out)` — once the caller has read the counts, all levels of all passes in one
    copy — cuts it to the exact-size Rulebook the consumers take."""
    __slots__ = ('subm', 'kvol', 'cap_in', 'cap_out', 'out_shape', 'indices', 'n_in_dev', 'n_out_dev', 'outids', 'nbr_out',
                 'nbr_in', 'indice_pairs', 'indice_num', 'args')

    def finish(self, n_in, n_out):
        """-> Rulebook with exact-size tensors; None when the capacity was too small (n_out > cap_out: rebuild the
        two-phase way)."""
        if n_out > self.cap_out or n_in > self.cap_in:
            return None
        rb = Rulebook()
        rb.subm, rb.kvol, rb.n_in, rb.n_out, rb.out_shape = self.subm, self.kvol, n_in, n_out, self.out_shape
        rb.indice_num = self.indice_num
        rb.outids = self.indices[:n_in] if self.subm else self.outids[:n_out]
        rb.nbr_out = _cut_table(self.nbr_out, self.kvol, self.cap_in if self.subm else self.cap_out, n_out)
        rb.nbr_in = None if self.subm else _cut_table(self.nbr_in, self.kvol, self.cap_in, n_in)
        rb.indice_pairs = None if self.indice_pairs is None else \
            _cut_table(self.indice_pairs, 2 * self.kvol, self.cap_in, n_in).view(self.kvol, 2, n_in)
        return rb


def _cut_table(table, rows, cap, n):
    """The first n columns of a (rows, cap) int32 table as a dense (rows, n) one: one pitched copy (dm_copy2d_f32 moves
    32-bit words, 16 bytes per lane when the pitches allow)."""
    if n == cap:
        return table.view(rows, cap)
    out = torch.empty((rows, n), dtype=torch.int32, device=table.device)
    if n > 0:
        _lib.check(_lib.lib().dm_copy2d_f32(_lib.ptr(table), cap, _lib.ptr(out), n, rows, n, _lib.stream()), 'dm_copy2d_f32')
    return out


def strided_capacity(cap_in, ksize, stride, first_cap):
    """Row capacity of a strided layer's output: an input touches at most prod(ceil(k / s)) outputs, but tables of
    that size would be mostly padding (KITTI: 29 k -> 34 k -> 19 k -> 8 k -> 6 k rows) — twice the capacity of the
    voxelizer bounds every level seen on LiDAR clouds, and a layer that still overflows is rebuilt two-phase
    (CapRulebook.finish -> None)."""
    worst = cap_in
    for k, s_ in zip(ksize, stride):
        worst *= -(-k // s_)
    return int(min(worst, max(2 * first_cap, 4096)))


def build_rulebook_cap(indices, n_dev, cap_in, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                       subm=False, with_pairs=True, cap_out=None, ws_tag='rulebook'):
    """build_rulebook with every size on the device: `indices` (cap_in, 4) of which the first *n_dev rows count.
    -> CapRulebook, or None when the capacity-sized entry does not apply (the caller builds two-phase)."""
    ksize, stride, padding, dilation = (_norm(v, 3) for v in (ksize, stride, padding, dilation))
    if indices.shape[1] != 4 or any(d != 1 for d in dilation) or cap_in <= 0:
        return None
    assert indices.dtype == torch.int32 and indices.is_contiguous() and indices.shape[0] >= cap_in
    _lib.require_device(indices)
    L = _lib.lib()
    dev = indices.device
    kvol = ksize[0] * ksize[1] * ksize[2]
    spatial = [int(s) for s in spatial_shape]
    c = CapRulebook()
    c.subm, c.kvol, c.cap_in, c.indices, c.n_in_dev = bool(subm), kvol, int(cap_in), indices, n_dev
    ws = _lib.workspace(L.dm_rulebook_workspace_bytes(cap_in, kvol), dev, ws_tag)
    c.indice_num = torch.empty((kvol,), dtype=torch.int32, device=dev)
    c.indice_pairs = torch.empty((kvol, 2, cap_in), dtype=torch.int32, device=dev) if with_pairs else None
    if subm:
        c.out_shape, c.cap_out, c.n_out_dev, c.outids, c.nbr_in = spatial, int(cap_in), n_dev, indices, None
        c.nbr_out = torch.empty((kvol, cap_in), dtype=torch.int32, device=dev)
        rc = L.dm_rulebook_subm_cap(_lib.ptr(indices), _lib.ptr(n_dev), int(cap_in), int(batch_size), _lib.ints(spatial),
                                    _lib.ints(ksize), _lib.ptr(c.nbr_out), _lib.ptr(c.indice_pairs), _lib.ptr(c.indice_num),
                                    _lib.ptr(ws), ws.numel(), _lib.stream())
    else:
        c.out_shape = get_conv_output_size(spatial, ksize, stride, padding, dilation)
        c.cap_out = int(cap_out)
        c.n_out_dev = torch.empty((1,), dtype=torch.int32, device=dev)
        c.outids = torch.empty((c.cap_out, 4), dtype=torch.int32, device=dev)
        c.nbr_out = torch.empty((kvol, c.cap_out), dtype=torch.int32, device=dev)
        c.nbr_in = torch.empty((kvol, cap_in), dtype=torch.int32, device=dev)
        rc = L.dm_rulebook_conv_cap(_lib.ptr(indices), _lib.ptr(n_dev), int(cap_in), int(batch_size), _lib.ints(spatial),
                                    _lib.ints(c.out_shape), _lib.ints(ksize), _lib.ints(stride), _lib.ints(padding),
                                    c.cap_out, _lib.ptr(c.n_out_dev), _lib.ptr(c.outids), _lib.ptr(c.nbr_out),
                                    _lib.ptr(c.nbr_in), _lib.ptr(c.indice_pairs), _lib.ptr(c.indice_num), _lib.ptr(ws),
                                    ws.numel(), _lib.stream())
    if rc == 2:           # DM_ERR_WORKSPACE: the occupancy bitmap does not fit — not an error, the two-phase path takes it
        return None
    _lib.check(rc, 'dm_rulebook_%s_cap' % ('subm' if subm else 'conv'))
    c.args = (batch_size, spatial, ksize, stride, padding, dilation, subm, with_pairs)
    return c


def get_indice_pairs_steps(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                           out_padding=0, subm=False, transpose=False, grid=None, ws_tag='rulebook'):
    """get_indice_pairs as a generator (see build_rulebook_steps)."""
    if transpose:
        raise NotImplementedError('transposed sparse conv is off the DetMatch hot path')
    rb = yield from build_rulebook_steps(indices, batch_size, spatial_shape, ksize, stride, padding,
                                         dilation, subm, ws_tag=ws_tag)
    rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
    return rb.outids, rb.indice_pairs, rb.indice_num


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None):
    """ops.py:46-105 -> (outids, indice_pairs (K,2,N), indice_pair_num (K)).

    The native gather tables ride along on the returned `indice_pairs` tensor
    (attribute `dm_tables`) so that indice_conv can use them without rebuilding."""
    if transpose:
        raise NotImplementedError('transposed sparse conv is off the DetMatch hot path')
    rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm)
    # (nbr_out, nbr_in, subm) — a plain tuple, so no reference cycle through the tensor
    rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
    return rb.outids, rb.indice_pairs, rb.indice_num


def _tables_for(indice_pairs, indice_pair_num, n_in, n_out, subm):
    tabs = getattr(indice_pairs, 'dm_tables', None)
    if tabs is not None:
        return tabs[0], (tabs[0] if tabs[2] else tabs[1])
    # foreign (reference-format) rulebook: rebuild the gather tables from the pair lists
    L = _lib.lib()
    kvol, _, stride = indice_pairs.shape
    dev = indice_pairs.device
    nbr_out = torch.empty((kvol, n_out), dtype=torch.int32, device=dev)
    _lib.check(L.dm_pairs_to_table(_lib.ptr(indice_pairs), _lib.ptr(indice_pair_num), kvol, stride,
                                   1, _lib.ptr(nbr_out), n_out, _lib.stream()), 'dm_pairs_to_table')
    if subm:
        return nbr_out, nbr_out
    nbr_in = torch.empty((kvol, n_in), dtype=torch.int32, device=dev)
    _lib.check(L.dm_pairs_to_table(_lib.ptr(indice_pairs), _lib.ptr(indice_pair_num), kvol, stride,
                                   0, _lib.ptr(nbr_in), n_in, _lib.stream()), 'dm_pairs_to_table')
    return nbr_out, nbr_in


# bench.py's roofline accounting: when a list is installed here, every gather-GEMM launch appends
# (c_in, c_out, rows_out, kvol, P) with P = number of rulebook pairs (one read-back per launch —
# measurement only, never enabled inside a timed region).  LAUNCH_TRACE_DIR records, in step with
# it, 'fwd' / 'dgrad' per gather-GEMM launch; LAUNCH_TRACE_W the weight-gradient launches as
# (c_in, c_out, kvol, P, n_in, n_out).
LAUNCH_TRACE = None
LAUNCH_TRACE_DIR = None
LAUNCH_TRACE_W = None


def tile_order(nbr):
    """Heavy-first launch order of a gather table's 16-row tiles (dm_spconv_tile_order), cached on
    the table tensor: one build per rulebook table, reused by every launch on it."""
    order = getattr(nbr, 'dm_tile_order', None)
    if order is None:
        L = _lib.lib()
        kvol, n = int(nbr.shape[0]), int(nbr.shape[1])
        order = torch.empty(((n + 15) // 16,), dtype=torch.int32, device=nbr.device)
        ws = _lib.workspace(L.dm_spconv_tile_order_workspace_bytes(), nbr.device, 'tile_order')
        _lib.check(L.dm_spconv_tile_order(_lib.ptr(nbr), n, kvol, _lib.ptr(order), _lib.ptr(ws),
                                          ws.numel(), _lib.stream()), 'dm_spconv_tile_order')
        nbr.dm_tile_order = order
    return order


# tables with fewer tiles than this are launch-latency sized: the order would cost more than it saves
TILE_ORDER_MIN_ROWS = 4096


PACK_ROWS = True       # (module switch of the equality tests and tools/spconv_pack_probe.py)


def packed_rows(nbr):
    """(nbr_packed, perm, tile order of nbr_packed) of a gather table (dm_spconv_pack_rows): rows grouped
    by equal neighbour mask, cached on the tensor (one build per rulebook, every launch reuses it)."""
    hit = getattr(nbr, 'dm_packed', None)
    if hit is None:
        L = _lib.lib()
        kvol, n = nbr.shape
        buf = torch.empty((n + (n + 15) // 16,), dtype=torch.int32, device=nbr.device)
        perm, order = buf[:n], buf[n:]
        packed = torch.empty_like(nbr)
        ws = _lib.workspace(L.dm_spconv_pack_rows_workspace_bytes(n), nbr.device, 'pack_rows')
        _lib.check(L.dm_spconv_pack_rows(nbr.data_ptr(), n, kvol, perm.data_ptr(), packed.data_ptr(),
                                         order.data_ptr(), ws.data_ptr(), ws.numel(), _lib.raw_stream()),
                   'dm_spconv_pack_rows')
        packed.dm_tile_order = order
        hit = nbr.dm_packed = (packed, perm, order)
    return hit


_STORAGE = {torch.float16: 1, torch.bfloat16: 2}      # DM_SP16_F16 / DM_SP16_BF16; 0 = DM_SP16_F32ROWS


def _gather_gemm(feat, filters, nbr, n_rows_out, cin, cout, transpose_w, flip_k):
    L = _lib.lib()
    kvol = nbr.shape[0]
    ci = cout if transpose_w else cin
    order = perm = None
    table = nbr
    if ci >= 32 and n_rows_out >= TILE_ORDER_MIN_ROWS:
        if PACK_ROWS:
            table, perm, order = packed_rows(nbr)
        else:
            order = tile_order(table)
    if LAUNCH_TRACE is not None:
        ci, co = (cout, cin) if transpose_w else (cin, cout)
        LAUNCH_TRACE.append((ci, co, int(n_rows_out), int(kvol), int((nbr >= 0).sum().item())))
        if LAUNCH_TRACE_DIR is not None:
            LAUNCH_TRACE_DIR.append('dgrad' if transpose_w else 'fwd')
    dev = feat.device
    out = torch.empty((n_rows_out, cin if transpose_w else cout), dtype=feat.dtype, device=dev)
    storage = _STORAGE.get(feat.dtype)
    if storage is None and min(cin, cout) >= 16:
        from .. import precision
        if precision.sparse_bf16():
            storage = 0
        elif precision.fp32_flavour() == 'fp32_split':
            storage = 3         # DM_SP16_F32SPLIT: fp32-class arithmetic on the bf16 instruction
    if storage is not None:
        # 16-bit matrix instructions (csrc/spconv16.hip): half-precision rows (indice_conv_half), or fp32 rows
        # with bf16 multiplicands in the mixed-precision mode
        if filters.dtype != feat.dtype:
            raise _lib.DetMatchHipError('sparse conv: features %s and filters %s differ' % (feat.dtype, filters.dtype))
        ws = _lib.workspace(L.dm_spconv16_workspace_bytes(kvol, cin, cout), dev, 'spconv16')
        rc = L.dm_spconv_gather_gemm16(_lib.ptr(feat), feat.shape[0], _lib.ptr(filters), storage, _lib.ptr(table),
                                       n_rows_out, kvol, cin, cout, int(transpose_w), int(flip_k), _lib.ptr(out),
                                       _lib.ptr(order), _lib.ptr(perm), _lib.ptr(ws), ws.numel(), _lib.stream())
        _lib.check(rc, 'dm_spconv_gather_gemm16')
        return out
    ws = _lib.workspace(L.dm_spconv_workspace_bytes(kvol, cin, cout), dev, 'spconv')
    rc = L.dm_spconv_gather_gemm(_lib.ptr(feat), feat.shape[0], _lib.ptr(filters), _lib.ptr(table),
                                 n_rows_out, kvol, cin, cout, int(transpose_w), int(flip_k),
                                 _lib.ptr(out), _lib.ptr(order), _lib.ptr(perm), _lib.ptr(ws), ws.numel(),
                                 _lib.stream())
    _lib.check(rc, 'dm_spconv_gather_gemm')
    return out


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out,
                inverse=False, subm=False):
    """ops.py:108-126 / spconv_ops.h:260-360: indice_conv_fp32 for fp32 tensors, indice_conv_half for
    torch.float16 (and, beyond the reference, torch.bfloat16) — the dtype dispatch of ops.py:111-126."""
    if inverse:
        raise NotImplementedError('inverse sparse conv is off the DetMatch hot path')
    if features.dtype not in (torch.float32, torch.float16, torch.bfloat16) or filters.dtype != features.dtype:
        raise NotImplementedError('sparse conv: fp32, fp16 or bf16 features and filters of one dtype')
    if features.dtype != torch.float32 and min(filters.shape[-2], filters.shape[-1]) < 16:
        raise NotImplementedError('half-precision sparse conv needs >= 16 channels on both sides')
    features = features.contiguous()
    filters = filters.contiguous()
    _lib.require_device(features, filters, indice_pairs)
    cin, cout = filters.shape[-2], filters.shape[-1]
    nbr_out, _ = _tables_for(indice_pairs, indice_pair_num, features.shape[0],
                             int(num_activate_out), subm)
    return _gather_gemm(features, filters, nbr_out, int(num_activate_out), cin, cout, 0, 0)


# ---- weight gradients of a backward pass in one launch pair (csrc/spconv.hip: dm_spconv_wgrad_batch) ------------
# Inside `with deferred_weight_grads():` the sparse-conv Functions hand their weight gradient to a queue instead of
# computing it: (saved input rows, output-row gradient, rulebook, the weight Parameter).  Leaving the context — on
# the thread that called backward(), after it has returned — flushes the queue: one rows launch + one reduce for
# all queued layers, results added straight into the parameters' .grad (allocated when absent).  Values are those
# of the per-layer calls (same chunks, same fixed-order sums).  The drivers that own a backward pass wrap it
# (SSL's early backward passes, OptimizerHook); a plain loss.backward() outside the context computes every weight
# gradient inside the pass as before.
WGRAD_BATCH = True     # (module switch of the equality tests)
_WGRAD_QUEUE = []
_WGRAD_DEPTH = [0]
WGRAD_BATCHES = [0]      # flushes so far (tests)


class deferred_weight_grads(object):
    def __enter__(self):
        _WGRAD_DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        _WGRAD_DEPTH[0] -= 1
        if _WGRAD_DEPTH[0] == 0:
            if exc[0] is None:
                _flush_weight_grads()
            else:
                del _WGRAD_QUEUE[:]
        return False


def defer_weight_grad(features, weight, out_bp, indice_pairs, indice_pair_num):
    """True: the layer's weight gradient was queued (the caller returns None for it)."""
    if not WGRAD_BATCH or _WGRAD_DEPTH[0] == 0 or features.dtype != torch.float32 or weight.dtype != torch.float32 or \
            not isinstance(weight, torch.nn.Parameter) or not weight.is_contiguous() or not features.is_cuda or \
            getattr(weight, '_post_accumulate_grad_hooks', None):
        # (a parameter somebody watches with a post-accumulate hook — FlatGradDDP in hooks mode — gets its
        # gradient inside the pass, where the hook expects it)
        return False
    cin, cout = weight.shape[-2], weight.shape[-1]
    if (cin, cout) not in ((4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128)) or indice_pairs.shape[2] == 0:
        return False
    if LAUNCH_TRACE_W is not None:
        LAUNCH_TRACE_W.append((cin, cout, int(indice_pairs.shape[0]), int(indice_pair_num.sum().item()),
                               int(features.shape[0]), int(out_bp.shape[0])))
    _WGRAD_QUEUE.append((features.contiguous(), weight, out_bp.contiguous(), indice_pairs, indice_pair_num,
                         torch.cuda.current_stream(features.device)))
    return True


def _flush_weight_grads():
    jobs = list(_WGRAD_QUEUE)
    del _WGRAD_QUEUE[:]
    if not jobs:
        return
    L = _lib.lib()
    dev = jobs[0][0].device
    # on the stream the pass's sparse layers ran on (the torch Stream object recorded at queue time — NOT an
    # ExternalStream around its raw handle: the caching allocator files allocations made under such a wrapper of
    # the default stream under a different stream id, and their blocks are then recycled out of stream order)
    with torch.no_grad(), torch.cuda.stream(jobs[0][5]):
        parts, seen = [[], []], set()
        for j in jobs:          # a weight queued twice (two passes in one backward): its second job adds to the first one's result
            first = j[1].grad is None and id(j[1]) not in seen
            seen.add(id(j[1]))
            parts[0 if first else 1].append(j)
        for accumulate, part in enumerate(parts):
            for lo in range(0, len(part), 16):
                chunk = part[lo:lo + 16]
                arr = (_lib.SpconvWgradJob * len(chunk))()
                outs = []
                # accumulate in place only if EVERY job of the chunk can (one flag per launch): otherwise each job
                # gets a fresh buffer that is added afterwards (a job writing into its own .grad with accumulate = 0
                # would lose the earlier pass and then be added to itself)
                in_place = bool(accumulate) and all(j[1].grad is not None and j[1].grad.is_contiguous() for j in chunk)
                for a, (feat, w, dy, pairs, num, _) in zip(arr, chunk):
                    if in_place:
                        tgt = out = w.grad
                    else:
                        tgt, out = None, torch.empty_like(w)
                    outs.append((w, out, tgt))
                    kvol, _, stride = pairs.shape
                    a.feat, a.out_grad, a.indice_pairs, a.indice_num = feat.data_ptr(), dy.data_ptr(), pairs.data_ptr(), num.data_ptr()
                    a.filt_grad = out.data_ptr()
                    a.pair_stride, a.kvol, a.cin, a.cout = int(stride), int(kvol), int(w.shape[-2]), int(w.shape[-1])
                acc_all = in_place
                ws = _lib.workspace(L.dm_spconv_wgrad_batch_workspace_bytes(arr, len(chunk)), dev, 'wgrad_batch')
                _lib.check(L.dm_spconv_wgrad_batch(arr, len(chunk), int(acc_all), _lib.ptr(ws), ws.numel(),
                                                   _lib.stream()), 'dm_spconv_wgrad_batch')
                for w, out, tgt in outs:
                    if w.grad is None:
                        w.grad = out
                    elif not acc_all:
                        w.grad.add_(out)
        WGRAD_BATCHES[0] += 1


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num,
                         inverse=False, subm=False, need_input_grad=True, defer_weight=None):
    """ops.py:142-158 / spconv_ops.h:363-456 -> [input_bp, filters_bp].  defer_weight: the weight Parameter when
    the caller is an autograd Function inside a backward pass — the weight gradient may then be queued for the
    pass's batched launch (defer_weight_grad) and filters_bp comes back as None."""
    if inverse:
        raise NotImplementedError('inverse sparse conv is off the DetMatch hot path')
    features = features.contiguous()
    filters = filters.contiguous()
    out_bp = out_bp.contiguous()
    _lib.require_device(features, filters, out_bp, indice_pairs)
    L = _lib.lib()
    dev = features.device
    cin, cout = filters.shape[-2], filters.shape[-1]
    kvol, _, stride = indice_pairs.shape
    n_in, n_out = features.shape[0], out_bp.shape[0]
    input_bp = None
    if need_input_grad:
        _, nbr_in = _tables_for(indice_pairs, indice_pair_num, n_in, n_out, subm)
        input_bp = _gather_gemm(out_bp, filters, nbr_in, n_in, cin, cout, 1, 1 if subm else 0)
    if defer_weight is not None and defer_weight_grad(features, defer_weight, out_bp, indice_pairs, indice_pair_num):
        return input_bp, None
    half = features.dtype if features.dtype != torch.float32 else None
    if half is not None:
        # indice_conv_backward_half: the weight gradient accumulates over all pairs — computed by the fp32
        # kernel on widened rows and rounded once (fp32 accumulation, as the gather-GEMMs)
        features, out_bp, filters = features.float(), out_bp.float(), filters.float()
    filters_bp = torch.empty_like(filters)
    if LAUNCH_TRACE_W is not None:
        LAUNCH_TRACE_W.append((cin, cout, int(kvol), int(indice_pair_num.sum().item()), int(n_in), int(n_out)))
    ws = _lib.workspace(L.dm_spconv_wgrad_workspace_bytes(stride, kvol, cin, cout), dev, 'wgrad')
    rc = L.dm_spconv_wgrad(_lib.ptr(features), _lib.ptr(out_bp), _lib.ptr(indice_pairs),
                           _lib.ptr(indice_pair_num), stride, kvol, cin, cout,
                           _lib.ptr(filters_bp), _lib.ptr(ws), ws.numel(), _lib.stream())
    _lib.check(rc, 'dm_spconv_wgrad')
    return input_bp, (filters_bp.to(half) if half is not None else filters_bp)
