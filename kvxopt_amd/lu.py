"""Sparse LU objects over the C ABI (kvx_lu_* in include/kvxhip.h): the numeric phase runs on the GPU only.

`LuSymbolic` / `LuNumeric` are what `kvxopt_amd.klu` wraps into the reference's opaque factors
(src/C/klu.c:276-279, 336-339).
"""
import ctypes

import numpy as np

from ._lib import lib, raise_for, pi, pd, as_i64, as_f64, i64, i64p, f64p, vp


def _take(ptr, count, dtype):
    """Copy a malloc'ed C array into numpy and free it (kvx_free)."""
    out = np.empty(count, dtype=dtype)
    if count:
        ctypes.memmove(out.ctypes.data, ptr, out.nbytes)
    lib().kvx_free(ptr)
    return out


class LuSymbolic:
    def __init__(self, n, colptr, rowind, values=None):
        self.n = int(n)
        self.colptr = as_i64(colptr)
        self.rowind = as_i64(rowind)
        v = None if values is None else as_f64(values)
        h = vp()
        rc = lib().kvx_lu_analyze(self.n, pi(self.colptr), pi(self.rowind), None if v is None else pd(v), ctypes.byref(h))
        raise_for(rc, "symbolic factorization failed")
        self._h = h

    def info(self):
        a = np.zeros(8, dtype=np.int64)
        raise_for(lib().kvx_lu_sym_info(self._h, pi(a)))
        keys = ("n", "nnz", "nsuper", "merges", "structurally_singular", "lnz_sym", "nlevels", "max_front")
        return dict(zip(keys, (int(x) for x in a)))

    def btf(self):
        """(number of diagonal blocks, number of block levels, block of every column)."""
        nb, nl = i64(), i64()
        blk = np.empty(self.n, dtype=np.int64)
        raise_for(lib().kvx_lu_sym_btf(self._h, ctypes.byref(nb), ctypes.byref(nl), pi(blk)))
        return nb.value, nl.value, blk

    def matching(self):
        r = np.empty(self.n, dtype=np.int64)
        raise_for(lib().kvx_lu_sym_matching(self._h, pi(r)))
        return r

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().kvx_lu_free_symbolic(h)
            except Exception:
                pass
            self._h = None


class LuNumeric:
    """numeric(A, Fs): factor on the GPU.  Keeps a reference to the symbolic object (the C side needs it alive)."""

    def __init__(self, sym, values):
        self.sym = sym
        v = as_f64(values).reshape(-1)
        h = vp()
        rc = lib().kvx_lu_factor(sym._h, v.size, pd(v), ctypes.byref(h))
        raise_for(rc, "factorization failed")
        self._h = h
        self.n = sym.n

    def refactor(self, values):
        v = as_f64(values).reshape(-1)
        raise_for(lib().kvx_lu_refactor(self._h, v.size, pd(v)))

    def refactor_dev(self, values_ptr, nnz):
        raise_for(lib().kvx_lu_refactor_dev(self._h, int(nnz), values_ptr))

    def info(self):
        a = np.zeros(8, dtype=np.int64)
        raise_for(lib().kvx_lu_num_info(self._h, pi(a)))
        keys = ("nfront", "nlevels", "max_front", "max_pivot_block", "panel_doubles", "arena_doubles", "passes", "factored")
        return dict(zip(keys, (int(x) for x in a)))

    def work(self):
        """flops, panel entries, update-matrix entries of one numeric factorisation; blocked fronts and their flops (kvx_lu_num_work)."""
        w = np.zeros(5)
        raise_for(lib().kvx_lu_num_work(self._h, pd(w)))
        return dict(zip(("flops", "panel_entries", "update_entries", "blocked_fronts", "blocked_flops"), (float(x) for x in w)))

    def graph_replays(self):
        """Replays of captured launch sequences (refactorisations and solves of the steady state) by this factor."""
        a = np.zeros(1, dtype=np.int64)
        raise_for(lib().kvx_lu_num_graph_replays(self._h, pi(a)))
        return int(a[0])

    def solve(self, B, trans="N", nrhs=None, ldB=None, offset=0):
        """B: 1-D float64 buffer holding an n x nrhs column-major block at `offset` with leading dimension ldB."""
        n = self.n
        if nrhs is None:
            nrhs = B.size // max(n, 1)
        if ldB is None:
            ldB = max(1, n)
        ptr = ctypes.cast(B.ctypes.data + 8 * offset, f64p)
        raise_for(lib().kvx_lu_solve(self._h, 0 if trans == "N" else 1, ptr, nrhs, ldB))

    def solve_dev(self, B_ptr, trans="N", nrhs=1, ldB=None):
        raise_for(lib().kvx_lu_solve_dev(self._h, 0 if trans == "N" else 1, B_ptr, nrhs, ldB or max(1, self.n)))

    def extract(self):
        n = self.n
        cnt = [i64() for _ in range(3)]
        ptrs_i = [i64p() for _ in range(6)]
        ptrs_x = [f64p() for _ in range(3)]
        P = np.empty(n, dtype=np.int64)
        Q = np.empty(n, dtype=np.int64)
        Rs = np.empty(n, dtype=np.float64)
        nb = i64()
        rp = i64p()
        rc = lib().kvx_lu_extract(self._h,
                                  ctypes.byref(cnt[0]), ctypes.byref(ptrs_i[0]), ctypes.byref(ptrs_i[1]), ctypes.byref(ptrs_x[0]),
                                  ctypes.byref(cnt[1]), ctypes.byref(ptrs_i[2]), ctypes.byref(ptrs_i[3]), ctypes.byref(ptrs_x[1]),
                                  ctypes.byref(cnt[2]), ctypes.byref(ptrs_i[4]), ctypes.byref(ptrs_i[5]), ctypes.byref(ptrs_x[2]),
                                  pi(P), pi(Q), pd(Rs), ctypes.byref(nb), ctypes.byref(rp))
        raise_for(rc)
        out = {}
        for name, c, a, b, x in (("L", cnt[0], 0, 1, 0), ("U", cnt[1], 2, 3, 1), ("F", cnt[2], 4, 5, 2)):
            out[name] = (_take(ptrs_i[a], n + 1, np.int64), _take(ptrs_i[b], c.value, np.int64), _take(ptrs_x[x], c.value, np.float64))
        out["P"], out["Q"], out["Rs"] = P, Q, Rs
        out["r"] = _take(rp, nb.value + 1, np.int64)
        return out

    def det(self):
        d = ctypes.c_double()
        raise_for(lib().kvx_lu_det(self._h, ctypes.byref(d)))
        return d.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().kvx_lu_free_numeric(h)
            except Exception:
                pass
            self._h = None
