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
