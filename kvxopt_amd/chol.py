"""numpy-level handle on the HIP supernodal Cholesky factor (C ABI: include/kvxhip.h).

This is the object behind the opaque factor `kvxopt_amd.cholmod.symbolic` returns
(the reference returns a PyCapsule around a cholmod_factor, cholmod.c:287-290).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import CholInfo, CholOpts, as_f64, as_i64, lib, pd, pi, raise_for


class Factor:
    def __init__(self, n, colptr, rowind, uplo="L", perm=None, opts=None):
        L = lib()
        self.n = int(n)
        self.colptr = as_i64(colptr)
        self.rowind = as_i64(rowind)
        self.uplo = uplo
        if uplo not in ("L", "U"):
            raise ValueError("possible values of uplo are: 'L', 'U'")
        o = CholOpts()
        L.kvx_chol_default_opts(ctypes.byref(o))
        for k, v in (opts or {}).items():
            if k == "nd_leaf":
                o.reserved[0] = int(v)
            elif k == "leaf_cols":          # 0 / negative = leaf-subtree amalgamation off
                o.reserved[1] = int(v) if int(v) > 0 else -1
            elif k == "leaf_rows":
                o.reserved[2] = int(v)
            elif k == "compare_given":      # a given perm is one candidate among the library's own orderings; least fill wins
                o.reserved[4] = 1 if v else 0
            elif k == "amd_auto_max":
                o.reserved[5] = int(v)
            elif k == "nd_min_n":          # ordering 0: smallest order for which the dissection is computed beside the minimum degree (-1: always)
                o.reserved[6] = int(v) if int(v) > 0 else -1
            elif k == "dbound_drop":        # with dbound > 0: a pivot below dbound^2 becomes 1e128 (its row drops out of the solves)
                o.reserved[3] = 1 if v else 0
            else:
                setattr(o, k, v)
        p = None
        if perm is not None:
            p = as_i64(perm)
            if p.size != self.n:
                raise TypeError("length of p is too small")
        h = ctypes.c_void_p()
        rc = L.kvx_chol_analyze(self.n, pi(self.colptr), pi(self.rowind), ord(uplo),
                                None if p is None else pi(p), ctypes.byref(o), ctypes.byref(h))
        raise_for(rc, "symbolic factorization failed")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().kvx_chol_free(h)
            except Exception:
                pass
            self._h = None

    # -- introspection ---------------------------------------------------------------------
    def info(self):
        inf = CholInfo()
        raise_for(lib().kvx_chol_get_info(self._h, ctypes.byref(inf)))
        return {f[0]: getattr(inf, f[0]) for f in CholInfo._fields_ if f[0] != "reserved"}

    def perm(self):
        p = np.empty(self.n, dtype=np.int64)
        raise_for(lib().kvx_chol_get_perm(self._h, pi(p)))
        return p

    def supernodes(self):
        ns = self.info()["nsuper"]
        sup = np.empty(ns + 1, dtype=np.int64)
        nrows = np.empty(ns, dtype=np.int64)
        parent = np.empty(ns, dtype=np.int64)
        level = np.empty(ns, dtype=np.int64)
        raise_for(lib().kvx_chol_get_supernodes(self._h, pi(sup), pi(nrows), pi(parent), pi(level)))
        return sup, nrows, parent, level

    def front_rows(self):
        """(rowptr, rowidx): sorted permuted row indices of every front, pivot rows first."""
        ns = self.info()["nsuper"]
        rp = np.zeros(ns + 1, dtype=np.int64)
        raise_for(lib().kvx_chol_get_front_rows(self._h, pi(rp), None))
        ri = np.zeros(max(int(rp[-1]), 1), dtype=np.int64)
        raise_for(lib().kvx_chol_get_front_rows(self._h, pi(rp), pi(ri)))
        return rp, ri[:int(rp[-1])]

    def timing(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        raise_for(lib().kvx_chol_last_timing(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def last_fused_path(self):
        """1: the last factorize_solve* call was one enqueue; 2: two enqueues (kvxhip.h kvx_chol_last_fused_path)."""
        return int(lib().kvx_chol_last_fused_path(self._h))

    FAMILIES = ("scatter_a", "front_small", "assemble_big", "potrf_diag", "trsm_panel", "syrk_trailing",
                "fwd_level", "bwd_level")

    def prof_select(self, family):
        fam = -1 if family is None else (self.FAMILIES.index(family) if isinstance(family, str) else int(family))
        raise_for(lib().kvx_chol_prof_select(self._h, fam))

    def prof_read(self):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        raise_for(lib().kvx_chol_prof_read(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # -- numeric -----------------------------------------------------------------------------
    def factorize(self, values):
        """values: host float64 array aligned with (colptr,rowind).  Raises
        ArithmeticError(minor) when the matrix is not positive definite."""
        v = as_f64(values)
        if v.size != (self.colptr[-1] if self.n else 0):
            raise TypeError("values do not match the analysed pattern")
        minor = ctypes.c_int64()
        rc = lib().kvx_chol_factorize(self._h, pd(v), ctypes.byref(minor))
        if rc == _lib.KVX_ENOTPOSDEF:
            raise ArithmeticError(int(minor.value))
        raise_for(rc, "factorization failed")

    def factorize_dev(self, values_ptr, sync=True):
        minor = ctypes.c_int64()
        if sync:
            rc = lib().kvx_chol_factorize_dev(self._h, values_ptr, ctypes.byref(minor))
            if rc == _lib.KVX_ENOTPOSDEF:
                raise ArithmeticError(int(minor.value))
        else:
            rc = lib().kvx_chol_factorize_async_dev(self._h, values_ptr)
        raise_for(rc, "factorization failed")

    def factorize_solve(self, values, B, nrhs=None, ldB=None, offset=0):
        """Numeric factorisation + solve A X = B with host buffers in one call (kvx_chol_factorize_solve): B overwritten."""
        v = as_f64(values)
        if v.size != (self.colptr[-1] if self.n else 0):
            raise TypeError("values do not match the analysed pattern")
        flat = B.reshape(-1, order="F") if B.ndim > 1 else B
        if nrhs is None:
            nrhs = 1 if B.ndim == 1 else B.shape[1]
        ptr = ctypes.cast(flat.ctypes.data + 8 * offset, _lib.f64p)
        minor = ctypes.c_int64()
        rc = lib().kvx_chol_factorize_solve(self._h, pd(v), ptr, int(nrhs), int(ldB or max(1, self.n)), ctypes.byref(minor))
        if rc == _lib.KVX_ENOTPOSDEF:
            raise ArithmeticError(int(minor.value))
        raise_for(rc, "factorization failed")
        return B

    def factorize_solve_async_dev(self, values_ptr, B_ptr, nrhs=1, ldB=None):
        """The same, enqueued only: later null-stream work is ordered behind it; status() reports the factorisation afterwards."""
        raise_for(lib().kvx_chol_factorize_solve_async_dev(self._h, values_ptr, B_ptr, int(nrhs), int(ldB or max(1, self.n))), "factorization failed")

    def factorize_solve_dev(self, values_ptr, B_ptr, nrhs=1, ldB=None):
        """Numeric factorisation + solve of A X = B in one enqueue on device buffers (the forward sweep pipelined behind the
        factorisation level by level); synchronises.  ArithmeticError(minor) when the matrix is not positive definite."""
        minor = ctypes.c_int64()
        rc = lib().kvx_chol_factorize_solve_dev(self._h, values_ptr, B_ptr, int(nrhs), int(ldB or max(1, self.n)), ctypes.byref(minor))
        if rc == _lib.KVX_ENOTPOSDEF:
            raise ArithmeticError(int(minor.value))
        raise_for(rc, "factorization failed")

    def status(self):
        minor = ctypes.c_int64()
        rc = lib().kvx_chol_status(self._h, ctypes.byref(minor))
        if rc == _lib.KVX_ENOTPOSDEF:
            raise ArithmeticError(int(minor.value))
        raise_for(rc)

    def solve(self, B, sys=0, nrhs=None, ldB=None, offset=0):
        """B: host float64 buffer (1-D view of a column-major n x nrhs block), overwritten."""
        flat = B.reshape(-1, order="F") if B.ndim > 1 else B
        if not flat.flags.c_contiguous and not flat.flags.f_contiguous:
            raise TypeError("B must be contiguous")
        if nrhs is None:
            nrhs = 1 if B.ndim == 1 else B.shape[1]
        if ldB is None:
            ldB = max(1, self.n)
        ptr = ctypes.cast(flat.ctypes.data + 8 * offset, _lib.f64p)
        raise_for(lib().kvx_chol_solve(self._h, int(sys), ptr, int(nrhs), int(ldB)), "solve step failed")
        return B

    def solve_dev(self, B_ptr, sys=0, nrhs=1, ldB=None, sync=True):
        """sync=False: enqueue only (kvx_chol_solve_async_dev); later null-stream work is ordered behind the solve."""
        fn = lib().kvx_chol_solve_dev if sync else lib().kvx_chol_solve_async_dev
        raise_for(fn(self._h, int(sys), B_ptr, int(nrhs), int(ldB or max(1, self.n))), "solve step failed")

    def spsolve(self, ncol, Bp, Bi, Bx, sys=0):
        Bp, Bi, Bx = as_i64(Bp), as_i64(Bi), as_f64(Bx)
        xp, xi, xx = _lib.i64p(), _lib.i64p(), _lib.f64p()
        rc = lib().kvx_chol_spsolve(self._h, int(sys), int(ncol), pi(Bp), pi(Bi), pd(Bx),
                                    ctypes.byref(xp), ctypes.byref(xi), ctypes.byref(xx))
        raise_for(rc, "solve step failed")
        try:
            Xp = np.ctypeslib.as_array(xp, shape=(ncol + 1,)).copy()
            nz = int(Xp[-1])
            Xi = np.ctypeslib.as_array(xi, shape=(max(nz, 1),))[:nz].copy()
            Xx = np.ctypeslib.as_array(xx, shape=(max(nz, 1),))[:nz].copy()
        finally:
            lib().kvx_free(xp)
            lib().kvx_free(xi)
            lib().kvx_free(xx)
        return Xp, Xi, Xx

    def diag(self):
        d = np.empty(self.n)
        raise_for(lib().kvx_chol_diag(self._h, pd(d)))
        return d

    def get_factor(self):
        lnz = ctypes.c_int64()
        raise_for(lib().kvx_chol_get_factor(self._h, ctypes.byref(lnz), None, None, None))
        Lp = np.empty(self.n + 1, dtype=np.int64)
        Li = np.empty(max(lnz.value, 1), dtype=np.int64)
        Lx = np.empty(max(lnz.value, 1))
        raise_for(lib().kvx_chol_get_factor(self._h, ctypes.byref(lnz), pi(Lp), pi(Li), pd(Lx)))
        return Lp, Li[:lnz.value], Lx[:lnz.value]
