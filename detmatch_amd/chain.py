"""Chain-level issue — host side of csrc/chain.hip (`dm_chain_run`).

A `Program` is the op table of one shape-static sub-graph of the training step (a BEV block, the 2D trunk, a
set-abstraction scale ...): every op is one launch-entry of libdetmatch_hip.so with its arguments given either as
immediates or as references into a small SLOT table (arena base addresses, the stream, data-dependent counts).
The table is built ONCE per shape signature by ordinary Python code that mirrors the module's forward (or
backward) — `prog.call('dm_dconv_gemm_planes', x, w, ...)` appends an op instead of launching it — and executed
with ONE ctypes call per pass: the kernels, their order and their arguments are those of the op-by-op path (results
are bit-identical to it: tests/test_chain_gpu.py), the per-launch interpreter / autograd / marshalling work is gone.

Reference for what the chains replace: the module-by-module issue of
pcdet/models/backbones_2d/base_bev_backbone.py:38-69, mmdet ResNet / FPN / RPNHead as configured at
configs/detmatch/001/detmatch/split_0.py:39-99, pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:58-104.

`DM_CHAIN=0` in the environment switches every chain off (the op-by-op path runs instead): the A/B switch of the
equality tests and of profiles/r05_ab_chain.txt.
"""
import ctypes
import os
import struct

import torch

from . import _lib

ENABLED = os.environ.get('DM_CHAIN', '1') == '1'
MAX_ARGS = 32
# families switched off individually (A/B, bisection): DM_CHAIN_OFF=bev,trunk2d,sa,sparse
OFF = set(f for f in os.environ.get('DM_CHAIN_OFF', '').split(',') if f)


def on(family, train=True):
    """Is the chained issue of this family (in this mode) switched on?"""
    return ENABLED and family not in OFF


# debugging aid (DM_CHAIN_DEBUG=1): every op of every chain is run on its own with an event behind it, so that
# `debug_report()` can say which op each stream is stuck at
DEBUG_EVENTS = None
if os.environ.get('DM_CHAIN_DEBUG'):
    import collections
    DEBUG_EVENTS = collections.deque(maxlen=40000)


def debug_report():
    """First incomplete op per stream (DM_CHAIN_DEBUG=1)."""
    seen = {}
    for name, i, op, stream, ev in DEBUG_EVENTS:
        if stream in seen:
            continue
        if not ev.query():
            seen[stream] = (name, i, op)
    return seen


class ChainOp(ctypes.Structure):
    """dm_chain_op (include/detmatch_hip.h)"""
    _fields_ = [('fn', ctypes.c_int32), ('nargs', ctypes.c_int32), ('slot', ctypes.c_int32 * MAX_ARGS),
                ('imm', ctypes.c_int64 * MAX_ARGS)]


class S(object):
    """Reference to a slot of the program's slot table (+ a constant byte / element offset)."""
    __slots__ = ('slot', 'off')

    def __init__(self, slot, off=0):
        self.slot, self.off = slot, off

    def __add__(self, n):
        return S(self.slot, self.off + int(n))

    def __repr__(self):
        return 'S(%d%+d)' % (self.slot, self.off)


_SIG_CACHE = {}


def _entry(name):
    """(function index, argument letters) of a launchable entry point; checks the compiled table against the
    ctypes signature table (a stale chain_tramp.inc must not go unnoticed)."""
    hit = _SIG_CACHE.get(name)
    if hit is None:
        L = _lib.lib()
        idx = L.dm_chain_fn_index(name.encode())
        if idx < 0:
            raise _lib.DetMatchHipError('%s cannot be called from a chain (not in csrc/chain_tramp.inc)' % name)
        sig = L.dm_chain_fn_signature(idx).decode()
        from_table = ''.join(_letter(a) for a in _lib.SIGNATURES[name][1])
        if sig != from_table:
            raise _lib.DetMatchHipError('chain trampoline of %s is stale (%s vs %s): run tools/gen_chain_tramp.py'
                                        % (name, sig, from_table))
        hit = _SIG_CACHE[name] = (idx, sig)
    return hit


def _letter(t):
    if t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
        return 'p'
    return {ctypes.c_int: 'i', ctypes.c_longlong: 'l', ctypes.c_size_t: 'z', ctypes.c_float: 'f',
            ctypes.c_double: 'd'}[t]


def f32_bits(v):
    return struct.unpack('<I', struct.pack('<f', float(v)))[0]


def f64_bits(v):
    return struct.unpack('<q', struct.pack('<d', float(v)))[0]


class Layout(object):
    """Bump allocator over one arena slot: `take(nbytes)` -> S(slot, offset), 256-byte aligned."""

    def __init__(self, slot):
        self.slot, self.size = slot, 0

    def take(self, nbytes, align=256):
        off = (self.size + align - 1) // align * align
        self.size = off + int(nbytes)
        return S(self.slot, off)

    def floats(self, *shape):
        n = 4
        for d in shape:
            n *= int(d)
        return self.take(n)


class Program(object):
    """One op table.  Slot 0 is always the HIP stream of the run."""

    STREAM = S(0)

    def __init__(self, name):
        self.name = name
        self.slot_names = ['stream']
        self.ops = []
        self.keep = []           # host arrays and device tensors whose addresses the table holds
        self.ws_bytes = 0        # scratch the ops need (one region: the ops of a chain run in stream order)
        self._table = None
        self._values = None
        self._failed = ctypes.c_int(-1)

    def slot(self, name):
        assert self._table is None, 'program already finalized'
        self.slot_names.append(name)
        return S(len(self.slot_names) - 1)

    def layout(self, name):
        return Layout(self.slot(name).slot)

    def need_workspace(self, nbytes):
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    def call(self, name, *args):
        assert self._table is None, 'program already finalized'
        idx, sig = _entry(name)
        if len(args) != len(sig):
            raise TypeError('%s takes %d arguments, got %d' % (name, len(sig), len(args)))
        enc = []
        for k, (a, t) in enumerate(zip(args, sig)):
            if isinstance(a, S):
                if t in 'fd':
                    raise TypeError('%s arg %d: a float cannot come from a slot' % (name, k))
                enc.append((a.slot, a.off))
            elif t == 'p':
                if a is None:
                    enc.append((-1, 0))
                elif isinstance(a, int):
                    enc.append((-1, a))
                elif isinstance(a, torch.Tensor):
                    if not a.is_cuda:
                        raise _lib.DetMatchHipError('%s arg %d: chains run on the MI355X only (got a %s tensor)'
                                                    % (name, k, a.device))
                    self.keep.append(a)
                    enc.append((-1, a.data_ptr()))
                elif isinstance(a, (ctypes.Array, ctypes.Structure)):
                    self.keep.append(a)
                    enc.append((-1, ctypes.addressof(a)))
                else:
                    raise TypeError('%s arg %d: cannot pass %r as a pointer' % (name, k, type(a)))
            elif t in 'ilz':
                enc.append((-1, int(a)))
            elif t == 'f':
                enc.append((-1, f32_bits(a)))
            else:
                enc.append((-1, f64_bits(a)))
        self.ops.append((idx, enc, name))

    def finalize(self):
        if self._table is None:
            tab = (ChainOp * max(len(self.ops), 1))()
            for i, (idx, enc, _) in enumerate(self.ops):
                op = tab[i]
                op.fn, op.nargs = idx, len(enc)
                for k, (s, imm) in enumerate(enc):
                    op.slot[k] = s
                    op.imm[k] = imm
            self._table = tab
            self._table_ptr = ctypes.addressof(tab)
            self._values = (ctypes.c_longlong * len(self.slot_names))()
            self._values_ptr = ctypes.addressof(self._values)
            self._run = _lib.lib().dm_chain_run
        return self

    def __len__(self):
        return len(self.ops)

    def run(self, values, stream=None):
        """values: one int per slot after the stream slot (same order as the `slot()` calls)."""
        if self._table is None:
            self.finalize()
        v = self._values
        v[0] = _lib.raw_stream() if stream is None else stream
        n = len(self.slot_names)
        if len(values) != n - 1:
            raise ValueError('%s: %d slot values expected, got %d' % (self.name, n - 1, len(values)))
        v[1:n] = values
        if DEBUG_EVENTS is not None:       # debugging aid: one op at a time with an event behind each
            sz = ctypes.sizeof(ChainOp)
            cur = torch.cuda.current_stream()
            for i in range(len(self.ops)):
                rc = self._run(self._table_ptr + i * sz, 1, self._values_ptr, n, self._failed)
                if rc != 0:
                    _lib.check(rc, 'chain %s, op %d (%s)' % (self.name, i, self.ops[i][2]))
                ev = torch.cuda.Event()
                ev.record(cur)
                DEBUG_EVENTS.append((self.name, i, self.ops[i][2], v[0], ev))
            return
        rc = self._run(self._table_ptr, len(self.ops), self._values_ptr, n, self._failed)
        if rc != 0:
            i = self._failed.value
            what = self.ops[i][2] if 0 <= i < len(self.ops) else '?'
            _lib.check(rc, 'chain %s, op %d (%s)' % (self.name, i, what))


# Weight-gradient half of a chain's backward on the side stream (_lib.aux_stream), underneath the rest of the backward pass
# on the main lane.  Scheduling only.  Off unless the driver of the iteration switches it on for its length
# (IterBasedSSLRunner.train: the gradients are then read by FlatGradDDP.collect, which waits for
# _lib.PENDING_GRAD_EVENTS first); a caller that reads `.grad` right after backward() never sees it on.
SIDE_WGRAD = [False]


def split_program(prog, names):
    """The table of `prog` (built, not necessarily finalized) in two halves sharing its slot layout: (ops NOT named in
    `names`, ops named in `names`), both finalized — or (None, None) when one half would be empty.  The caller guarantees
    that nothing of the first half reads what the second writes and that the second may run after the whole first."""
    picked = [op for op in prog.ops if op[2] in names]
    if not picked or len(picked) == len(prog.ops):
        return None, None
    halves = []
    for tag, keep in (('.a', False), ('.b', True)):
        half = Program(prog.name + tag)
        half.slot_names = list(prog.slot_names)
        half.ops = [op for op in prog.ops if (op[2] in names) == keep]
        half.keep, half.ws_bytes = prog.keep, prog.ws_bytes
        halves.append(half.finalize())
    return halves[0], halves[1]


def run_split(first, second, vals, ws_index, ws_bytes, device, tensors):
    """`first` on the current stream, `second` behind it on the side stream with a scratch region of its own
    (vals[ws_index]); `tensors`: everything the tables address (kept alive for the side stream).  The event behind the
    second half goes to _lib.PENDING_GRAD_EVENTS."""
    first.run(vals)
    main = torch.cuda.current_stream(device)
    ready = torch.cuda.Event()
    ready.record(main)
    side = _lib.aux_stream(device)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        ws2 = _lib.workspace(ws_bytes, device, 'chain') if ws_bytes else None
        vals2 = list(vals)
        vals2[ws_index] = 0 if ws2 is None else ws2.data_ptr()
        second.run(vals2)
        done = torch.cuda.Event()
        done.record(side)
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _lib.PENDING_GRAD_EVENTS.append(done)


def workspace(prog, device):
    """The scratch region of a program run on the current stream / thread (None if it needs none)."""
    if not prog.ws_bytes:
        return None
    return _lib.workspace(prog.ws_bytes, device, 'chain')


class DeviceTable(object):
    """A small table of fixed-size rows (numpy structured array) with a device copy, uploaded once."""

    def __init__(self, rows, device):
        import numpy as np
        self.host = rows
        self.dev = torch.from_numpy(rows.view(np.uint8).reshape(-1).copy()).to(device)
        self.n = len(rows)
