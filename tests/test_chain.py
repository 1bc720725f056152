"""Chain interpreter, host side (no GPU): the generated trampolines are current, the interpreter resolves names /
signatures, an op table marshals every argument class correctly and errors carry the failing op."""
import ctypes
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_trampolines_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_chain_tramp.py'), '--check'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_launch_entry_is_callable_from_a_chain():
    from detmatch_amd import _lib, chain
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_chain_tramp
    L = _lib.lib()
    fns = gen_chain_tramp.chain_functions(_lib.SIGNATURES)
    assert L.dm_chain_fn_count() == len(fns) >= 90
    for name, codes in fns:
        idx, sig = chain._entry(name)
        assert L.dm_chain_fn_name(idx).decode() == name and sig == codes
    assert L.dm_chain_fn_index(b'dm_chain_run') == -1          # the interpreter is not re-entrant by table
    assert L.dm_chain_fn_index(b'no_such_entry') == -1
    with pytest.raises(_lib.DetMatchHipError):
        chain._entry('dm_lap_host')


def test_program_marshals_slots_and_immediates():
    """dm_rowgemm_parts / size queries are not launchable, so the marshalling is checked on an entry that validates its
    arguments BEFORE touching the device: dm_fill_bytes(dst, value, 0 bytes) returns DM_OK without a launch,
    dm_copy2d_f32 with rows = 0 likewise, and a NULL destination is reported as the failing op."""
    from detmatch_amd import _lib, chain
    p = chain.Program('t')
    a = p.slot('a')
    n = p.slot('n')
    p.call('dm_fill_bytes', a + 16, 0, 0, chain.Program.STREAM)          # nbytes == 0: no launch
    p.call('dm_copy2d_f32', a, 4, a + 64, 4, n, 4, chain.Program.STREAM)  # rows from a slot (0): no launch
    p.call('dm_fill_bytes', None, 0, 64, chain.Program.STREAM)            # NULL dst with bytes: invalid argument
    assert len(p) == 3
    with pytest.raises(_lib.DetMatchHipError) as e:
        p.run([4096, 0], stream=0)
    assert 'op 2 (dm_fill_bytes)' in str(e.value)
    # encoding of the table itself
    t = p._table
    assert t[0].nargs == 4 and t[0].slot[0] == a.slot and t[0].imm[0] == 16 and t[0].slot[3] == 0
    assert t[1].slot[4] == n.slot and t[1].imm[4] == 0 and t[1].imm[1] == 4
    assert ctypes.sizeof(chain.ChainOp) == 8 + 4 * 32 + 8 * 32
    with pytest.raises(TypeError):
        p2 = chain.Program('u')
        p2.call('dm_fill_bytes', 0, 0, 0)      # wrong arity


def test_float_bits():
    from detmatch_amd import chain
    assert chain.f32_bits(1.0) == 0x3f800000 and chain.f32_bits(-2.0) == 0xc0000000
    assert chain.f64_bits(1.0) == 0x3ff0000000000000


def test_derived_weights_go_stale_when_any_watched_tensor_is_rewritten():
    """the fused optimizer reports the byte range it rewrote; a chain whose FIRST watched tensor is
    frozen (outside every such range) must still re-derive its packed weights when a later watched tensor is inside."""
    import torch
    from detmatch_amd import dense_chain, dense_conv

    class _Prog(object):
        runs = 0

        def run(self, slots):
            _Prog.runs += 1

    arena = torch.zeros(64)
    frozen, trained = torch.zeros(8), arena[16:32]            # frozen lives elsewhere, trained inside the arena
    for order in ((frozen, trained), (trained, frozen)):
        w = dense_chain._Weights(torch.device('cpu'))
        w.watch += list(order)
        w.prog = _Prog()
        _Prog.runs = 0
        w.refresh()
        assert _Prog.runs == 1                                # first use derives
        w.refresh()
        assert _Prog.runs == 1                                # nothing changed
        for i in range(5):                                    # five optimizer steps over the arena's bytes
            dense_conv.weights_changed(arena.data_ptr(), arena.data_ptr() + 4 * arena.numel())
            w.refresh()
            assert _Prog.runs == 2 + i, order
        # a rewrite somewhere else (the teacher's EMA range) leaves this chain alone
        other = torch.zeros(32)
        dense_conv.weights_changed(other.data_ptr(), other.data_ptr() + 4 * other.numel())
        w.refresh()
        assert _Prog.runs == 6
        dense_conv.weights_changed()                          # "everything"
        w.refresh()
        assert _Prog.runs == 7
    # the primitive: ranges are half-open, addresses ascending
    g = dense_conv._GENERATION[0]
    dense_conv.weights_changed(100, 200)
    assert dense_conv._stale_any(g, [50, 150, 300]) and dense_conv._stale_any(g, [100])
    assert not dense_conv._stale_any(g, [50, 200, 300]) and not dense_conv._stale_any(g, [])
    assert not dense_conv._stale_any(dense_conv._GENERATION[0], [150])


def test_chain_run_rejects_an_op_with_the_wrong_argument_count():
    """The C-ABI is public: a hand-built table whose nargs is shorter than the entry's signature must not reach the
    trampoline (it would read stale stack values as pointers)."""
    from detmatch_amd import _lib, chain
    L = _lib.lib()
    idx, sig = chain._entry('dm_fill_bytes')
    ops = (chain.ChainOp * 2)()
    for o in ops:
        o.fn, o.nargs = idx, len(sig)
        for k in range(len(sig)):
            o.slot[k], o.imm[k] = -1, 0                 # dst NULL, 0 bytes: DM_OK without a launch
    failed = ctypes.c_int(-5)
    assert L.dm_chain_run(ops, 2, None, 0, ctypes.byref(failed)) == 0 and failed.value == -1
    ops[1].nargs = len(sig) - 1
    rc = L.dm_chain_run(ops, 2, None, 0, ctypes.byref(failed))
    assert rc != 0 and failed.value == 1


def test_split_program_keeps_order_slots_and_shares_the_tables():
    """chain.split_program: two halves of one op table (the weight-gradient half of a backward chain goes to the side
    stream): every op lands in exactly one half, in its original order, both halves take the values of the full table."""
    from detmatch_amd import chain
    p = chain.Program('t')
    a, n = p.slot('a'), p.slot('n')
    p.call('dm_fill_bytes', a, 0, 0, chain.Program.STREAM)
    p.call('dm_copy2d_f32', a, 4, a + 64, 4, n, 4, chain.Program.STREAM)
    p.call('dm_fill_bytes', a + 8, 0, 0, chain.Program.STREAM)
    p.call('dm_copy2d_f32', a + 128, 4, a + 192, 4, n, 4, chain.Program.STREAM)
    first, second = chain.split_program(p, ('dm_copy2d_f32',))
    assert [o[2] for o in first.ops] == ['dm_fill_bytes'] * 2 and [o[2] for o in second.ops] == ['dm_copy2d_f32'] * 2
    assert first.ops[1][1][0] == (a.slot, 8) and second.ops[1][1][0] == (a.slot, 128)      # original order inside a half
    assert first.slot_names == second.slot_names == p.slot_names
    for half in (first, second):
        half.run([4096, 0], stream=0)              # zero rows / zero bytes: no launch, DM_OK
    assert chain.split_program(p, ('no_such_entry',)) == (None, None)
    assert chain.split_program(p, ('dm_fill_bytes', 'dm_copy2d_f32')) == (None, None)
    assert chain.SIDE_WGRAD == [False]               # only ever on for the length of an iteration (IterBasedSSLRunner.train)
