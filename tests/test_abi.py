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


def test_lu_abi_argument_checks_need_no_gpu():
    """Status codes of the kvx_lu_* entry points for bad arguments (mapped to the reference's exceptions by klu.py)."""
    L = _lib.lib()
    i64p, f64p, vp = _lib.i64p, _lib.f64p, _lib.vp
    cp = np.array([0, 1, 2], dtype=np.int64); ri = np.array([0, 1], dtype=np.int64); v = np.array([1.0, 2.0])
    h = vp()
    assert L.kvx_lu_analyze(0, _lib.pi(cp), _lib.pi(ri), _lib.pd(v), ctypes.byref(h)) == _lib.KVX_EINVAL          # klu.c:256-260
    bad = np.array([0, 2, 1], dtype=np.int64)
    assert L.kvx_lu_analyze(2, _lib.pi(bad), _lib.pi(ri), _lib.pd(v), ctypes.byref(h)) == _lib.KVX_EINVAL
    assert L.kvx_lu_analyze(2, _lib.pi(cp), _lib.pi(ri), _lib.pd(v), ctypes.byref(h)) == _lib.KVX_OK
    info = np.zeros(8, dtype=np.int64)
    assert L.kvx_lu_sym_info(h, _lib.pi(info)) == _lib.KVX_OK and info[0] == 2 and info[1] == 2 and info[4] == 0
    nb, nl = ctypes.c_int64(), ctypes.c_int64()
    assert L.kvx_lu_sym_btf(h, ctypes.byref(nb), ctypes.byref(nl), None) == _lib.KVX_OK and nb.value == 2 and nl.value == 1
    n_ = vp()
    assert L.kvx_lu_factor(h, 5, _lib.pd(v), ctypes.byref(n_)) == _lib.KVX_EINVAL                                  # other pattern
    if L.kvx_device_count() == 0:
        assert L.kvx_lu_factor(h, 2, _lib.pd(v), ctypes.byref(n_)) == _lib.KVX_EDEVICE                             # no CPU fallback
        assert b"no CPU fallback" in L.kvx_last_error()
    assert L.kvx_lu_solve(None, 0, _lib.pd(v), 1, 2) == _lib.KVX_EINVAL
    assert L.kvx_lu_det(None, None) == _lib.KVX_EINVAL
    L.kvx_lu_free_symbolic(h)
    L.kvx_lu_free_numeric(None)


def test_cone_block_operations_fail_loudly_without_gpu():
    """The 'q' / 's' block operations of kvxopt_amd.misc have no host arithmetic behind them: without a HIP device they raise
    (a vector without such blocks needs no device for the storage helpers -- plain copies -- and the argument checks of the C ABI
    answer on any box)."""
    if _lib.lib().kvx_device_count() > 0:
        pytest.skip("GPU present")
    from kvxopt_amd import misc
    from kvxopt_amd.base import matrix
    dims = {"l": 2, "q": [], "s": []}
    x, y = matrix(np.arange(4.0)), matrix(0.0, (6, 1))
    misc.pack(x, y, dims, 1, 1, 2)                       # mnl = 1: three leading entries copied, nothing else touched
    assert np.array_equal(y._a, [0.0, 0.0, 1.0, 2.0, 3.0, 0.0])
    misc.unpack(y, x, dims, 1, 2, 1)
    assert np.array_equal(x._a, [0.0, 1.0, 2.0, 3.0])
    dims_s = {"l": 0, "q": [], "s": [2]}
    for call in (lambda: misc.pack(matrix(np.ones(4)), matrix(0.0, (3, 1)), dims_s),
                 lambda: misc.compute_scaling(matrix(np.eye(2).reshape(-1)), matrix(np.eye(2).reshape(-1)), matrix(0.0, (2, 1)), dims_s),
                 lambda: misc.max_step(matrix(np.ones(4)), dims_s),
                 lambda: misc.trisc(matrix(np.ones(4)), dims_s)):
        with pytest.raises(RuntimeError):
            call()
    L = _lib.lib()
    assert L.kvx_nts_scale_dev(-1, None, None, None, None, 1, 1, 0, None, 0) == _lib.KVX_EINVAL
    assert L.kvx_nts_scale_dev(1, None, None, None, None, 1, 1, 2, None, 0) == _lib.KVX_EINVAL      # form is 0 or 1
    assert L.kvx_nts_pack_dev(0, None, None, None, None, None, 0) == _lib.KVX_OK                     # no blocks: nothing to do
    assert L.kvx_nts_prod_dev(1, None, None, None, None, 3, None) == _lib.KVX_EINVAL
