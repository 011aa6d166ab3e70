"""The C-ABI library loads on a CPU-only box and exports every symbol include/pylattice_hip.h declares; without a
GPU the product path fails loudly (no CPU fallback).  No compute calls here."""
import os
import re

import numpy as np
import pytest

from pylatticedso_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pylattice_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pl_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_capi.EXPORTS)


def test_library_exports_every_symbol():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _capi.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pl_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.PlError) as e:
        _capi.HipLattice(np.zeros((2, 3)), [[0, 1]], [0.1], [0, 1.0, 0], [0, 20, 0], 1013.0, 0.3)
    assert e.value.code == _capi.PL_ERR_NODEVICE
