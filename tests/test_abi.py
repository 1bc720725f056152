"""The C-ABI library loads and exports every symbol include/detmatch_hip.h declares
(no compute calls — this runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, 'include', 'detmatch_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(dm_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
    from detmatch_amd.csrc import build
    path = build.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), 'missing export %s' % n


def test_python_binding_covers_header():
    from detmatch_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    L = _lib.lib()
    assert b'gfx950' in L.dm_version()
    assert L.dm_error_string(0) == b'ok'
    # host-only size queries are safe without a GPU
    assert L.dm_rulebook_workspace_bytes(1000, 27) > 0
    assert L.dm_hard_voxelize_workspace_bytes(1000, 2) > 0
    assert L.dm_spconv_workspace_bytes(27, 64, 64) == 27 * 64 * 64 * 4


def test_integration_guide_names_every_entry_point():
    """INTEGRATION.md shows, for each C-ABI entry, the reference binding / behaviour it stands in for."""
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = [n for n in _declared() if n not in doc]
    assert not missing, missing


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from detmatch_amd import _lib, voxel
    with pytest.raises(_lib.DetMatchHipError):
        voxel.voxelize_batch([torch.zeros(10, 4)], [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1],
                             5, 100)
