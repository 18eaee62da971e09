"""The C-ABI library loads without a GPU and exports every symbol include/kvxhip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from kvxopt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "kvxhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kvx_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    L = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 40
    for name in decl:
        assert hasattr(L, name), "missing export: " + name
    assert set(decl) == set(_lib.exported_symbols())


def test_version_and_device_probe():
    L = _lib.lib()
    assert b"gfx950" in L.kvx_version()
    assert L.kvx_device_count() >= 0


def test_numeric_path_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device factorize must raise, not compute."""
    if _lib.lib().kvx_device_count() > 0:
        pytest.skip("GPU present")
    from kvxopt_amd.chol import Factor
    F = Factor(2, [0, 2, 3], [0, 1, 1])
    with pytest.raises(RuntimeError):
        F.factorize(np.array([2.0, 1.0, 2.0]))
    with pytest.raises(ValueError):
        F.solve(np.ones(2))          # symbolic factor (cholmod.c:452-453)
