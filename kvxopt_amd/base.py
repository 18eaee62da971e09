"""Host containers and sparse BLAS of the hot path, mirroring the subset of `kvxopt.base` that
`misc.kkt_chol2` and the LP/QP drivers use (reference: src/C/base.c, dense.c, sparse.c).

* `matrix`   : dense, column-major, typecode 'd', 'i' or 'z' (layout of src/C/kvxopt.h:48-56; 'z' = complex128, accepted by
  the linear solvers `cholmod` / `klu`, which solve complex systems through their real 2n x 2n embedding).
* `spmatrix` : compressed-column storage with sorted rows (kvxopt.h:58-69, doc/source/c-api.rst:121-179).
* `gemm(..., partial=True)`, `syrk(...)`, `gemv(...)` on sparse operands run on the GPU through the
  C ABI (kvx_atda_*, kvx_spmv_dev); there is no CPU arithmetic fallback for them.

The containers are deliberately small: they are the drop-in vocabulary for the parity tests and for
users without kvxopt installed, not a re-implementation of kvxopt's type system.  All functions also
accept kvxopt's own `matrix`/`spmatrix` objects (buffer protocol / `.CCS`).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, lib, raise_for


# --------------------------------------------------------------------------------------------
class matrix:
    """Dense column-major matrix (numpy-backed).  `matrix(x, size=None, tc='d')`."""

    def __init__(self, x=0.0, size=None, tc=None):
        if isinstance(x, matrix):
            arr = x._a.copy()
            size = size or x.size
        elif isinstance(x, np.ndarray):
            arr = np.asarray(x, order="F")
            if size is None:
                size = arr.shape if arr.ndim == 2 else (arr.size, 1)
            arr = arr.reshape(-1, order="F")
        elif np.isscalar(x):
            if size is None:
                size = (1, 1)
            arr = np.full(size[0] * size[1], x)
        else:
            x = list(x)
            if x and isinstance(x[0], (list, tuple)):           # list of columns, as kvxopt
                ncols = len(x)
                nrows = len(x[0])
                arr = np.array([v for col in x for v in col])
                size = size or (nrows, ncols)
            else:
                arr = np.array(x)
                size = size or (len(x), 1)
        if tc is None:
            tc = "i" if arr.dtype.kind in "iu" else ("z" if arr.dtype.kind == "c" else "d")
        if tc not in ("i", "d", "z"):
            raise TypeError("tc must be 'i', 'd' or 'z'")
        self.typecode = tc
        self._a = np.ascontiguousarray(arr, dtype={"i": np.int64, "d": np.float64, "z": np.complex128}[tc]).copy()
        self.size = (int(size[0]), int(size[1]))
        if self._a.size != self.size[0] * self.size[1]:
            raise TypeError("wrong number of elements")

    # numpy interop: a writable (nrows, ncols) Fortran-ordered view
    @property
    def a(self):
        return self._a.reshape(self.size, order="F")

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return self._a.size

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            r = self.a[idx]
            return matrix(r) if isinstance(r, np.ndarray) else self._scalar(r)
        r = self._a[idx]
        if isinstance(r, np.ndarray):
            return matrix(r, (r.size, 1), self.typecode)
        return self._scalar(r)

    def _scalar(self, r):
        return int(r) if self.typecode == "i" else (complex(r) if self.typecode == "z" else float(r))

    def __setitem__(self, idx, v):
        v = v._a if isinstance(v, matrix) and not isinstance(idx, tuple) else (v.a if isinstance(v, matrix) else v)
        if isinstance(idx, tuple):
            self.a[idx] = v
        else:
            self._a[idx] = v

    def __iter__(self):
        return iter(self._a.tolist())

    def _bin(self, o, f):
        o = o._a if isinstance(o, matrix) else o
        return matrix(f(self._a, o), self.size)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return matrix(np.subtract(o._a if isinstance(o, matrix) else o, self._a), self.size)
    def __neg__(self): return matrix(-self._a, self.size)
    def __pos__(self): return matrix(self._a.copy(), self.size)

    def __mul__(self, o):
        if isinstance(o, matrix):
            return matrix(self.a @ o.a)
        return matrix(self._a * o, self.size)

    def __rmul__(self, o):
        return matrix(self._a * o, self.size)

    def __truediv__(self, o):
        return matrix(self._a / o, self.size)

    def __pow__(self, p):
        return matrix(self._a ** p, self.size)

    @property
    def T(self):
        return matrix(np.asfortranarray(self.a.T))

    def __repr__(self):
        return "<%dx%d matrix, tc='%s'>" % (self.size[0], self.size[1], self.typecode)


def _as_ccs(A):
    """(nrows, ncols, colptr, rowind, values) as numpy arrays from our spmatrix or kvxopt's."""
    if isinstance(A, spmatrix):
        return A.size[0], A.size[1], A.colptr, A.rowind, A.values
    cp, ri, v = A.CCS
    v = np.array(v).reshape(-1)
    return (A.size[0], A.size[1], np.array(cp, dtype=np.int64).reshape(-1),
            np.array(ri, dtype=np.int64).reshape(-1), v.astype(np.complex128 if v.dtype.kind == "c" else np.float64))


class spmatrix:
    """Sparse matrix in compressed-column storage.  `spmatrix(V, I, J, size=None, tc='d')` as in
    kvxopt (sparse.c:2587 SpMatrix_NewFromIJV): duplicate (i, j) entries are summed, rows are sorted."""

    def __init__(self, V, I, J, size=None, tc="d"):
        I = np.asarray(list(I) if not isinstance(I, (np.ndarray, matrix)) else (I._a if isinstance(I, matrix) else I), dtype=np.int64).reshape(-1)
        J = np.asarray(list(J) if not isinstance(J, (np.ndarray, matrix)) else (J._a if isinstance(J, matrix) else J), dtype=np.int64).reshape(-1)
        if isinstance(V, matrix):
            V = V._a
        if np.isscalar(V):
            V = np.full(I.size, V)
        V = np.asarray(V).reshape(-1)
        if tc == "d" and V.dtype.kind == "c":
            tc = "z"
        vdt = np.complex128 if tc == "z" else np.float64
        V = V.astype(vdt)
        if not (V.size == I.size == J.size):
            raise TypeError("V, I, J must have the same length")
        if size is None:
            size = (int(I.max()) + 1 if I.size else 0, int(J.max()) + 1 if J.size else 0)
        m, n = int(size[0]), int(size[1])
        if I.size and (I.min() < 0 or I.max() >= m or J.min() < 0 or J.max() >= n):
            raise TypeError("index out of range")
        self.size = (m, n)
        self.typecode = tc
        order = np.lexsort((I, J))
        I, J, V = I[order], J[order], V[order]
        if I.size:
            key = J * max(m, 1) + I
            uniq, start = np.unique(key, return_index=True)
            V = np.add.reduceat(V, start)
            I = I[start]
            J = J[start]
        self.colptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(self.colptr, J + 1, 1)
        np.cumsum(self.colptr, out=self.colptr)
        self.rowind = I.copy()
        self.values = V.astype(vdt).copy()

    @classmethod
    def from_ccs(cls, m, n, colptr, rowind, values):
        S = cls.__new__(cls)
        S.size = (int(m), int(n))
        values = np.asarray(values)
        S.typecode = "z" if values.dtype.kind == "c" else "d"
        S.colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        S.rowind = np.ascontiguousarray(rowind, dtype=np.int64)
        S.values = np.ascontiguousarray(values, dtype=np.complex128 if S.typecode == "z" else np.float64)
        return S

    @property
    def V(self):
        return matrix(self.values.copy(), (self.values.size, 1))

    @property
    def I(self):
        return matrix(self.rowind.copy(), (self.rowind.size, 1), "i")

    @property
    def J(self):
        return matrix(np.repeat(np.arange(self.size[1], dtype=np.int64), np.diff(self.colptr)), None, "i")

    @property
    def CCS(self):
        return (matrix(self.colptr.copy(), None, "i"), matrix(self.rowind.copy(), None, "i"), matrix(self.values.copy()))

    def __len__(self):
        return self.values.size

    @property
    def T(self):
        m, n = self.size
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(self.colptr))
        return spmatrix(self.values, cols, self.rowind, (n, m), self.typecode)

    def __iadd__(self, o):
        """S += H on the union pattern (sparse.c:4299 spmatrix_iadd)."""
        om, on, ocp, ori, ov = _as_ccs(o)
        if (om, on) != self.size:
            raise TypeError("incompatible dimensions")
        cols = np.repeat(np.arange(self.size[1], dtype=np.int64), np.diff(self.colptr))
        ocols = np.repeat(np.arange(on, dtype=np.int64), np.diff(ocp))
        R = spmatrix(np.concatenate([self.values, ov]), np.concatenate([self.rowind, ori]),
                     np.concatenate([cols, ocols]), self.size)
        self.colptr, self.rowind, self.values = R.colptr, R.rowind, R.values
        return self

    def todense(self):
        D = np.zeros(self.size, dtype=self.values.dtype)
        cols = np.repeat(np.arange(self.size[1], dtype=np.int64), np.diff(self.colptr))
        D[self.rowind, cols] = self.values
        return D

    def __repr__(self):
        return "<%dx%d sparse matrix, tc='%s', nnz=%d>" % (self.size[0], self.size[1], self.typecode, self.values.size)


def spdiag(x):
    """Sparse diagonal matrix from a dense vector (subset of kvxopt.spdiag)."""
    v = x._a if isinstance(x, matrix) else np.asarray(x, dtype=np.float64).reshape(-1)
    n = v.size
    return spmatrix(v, np.arange(n), np.arange(n), (n, n))


def _dense_buffer(B, tc="d"):
    """Writable 1-D numpy view (float64, or complex128 for tc = 'z') of a dense matrix (ours or kvxopt's), column-major."""
    want = np.complex128 if tc == "z" else np.float64
    if isinstance(B, matrix):
        if B.typecode != tc:
            raise TypeError("B must be a '%s' matrix" % tc)
        return B._a, B.size
    if isinstance(B, np.ndarray):
        if B.dtype != want:
            raise TypeError("B must be %s" % np.dtype(want).name)
        if B.ndim == 1:
            return B, (B.size, 1)
        if not B.flags.f_contiguous:
            raise TypeError("B must be column-major")
        return B.reshape(-1, order="F"), B.shape
    mv = memoryview(B)                                            # kvxopt matrix: buffer protocol (dense.c:1350-1385)
    if mv.format != ("Zd" if tc == "z" else "d"):
        raise TypeError("B must be a '%s' matrix" % tc)
    arr = np.asarray(mv)
    size = tuple(B.size)
    return arr.reshape(-1, order="F") if arr.ndim == 2 else arr, size


# --------------------------------------------------------------------------------------------
# sparse BLAS on the GPU
_plans = {}


def _atda_plan(ml, n, Gp, Gi):
    key = (ml, n, _lib.pattern_digest(Gp, Gi))
    P = _plans.get(key)
    if P is None:
        h = ctypes.c_void_p()
        raise_for(lib().kvx_atda_plan(ml, n, _lib.pi(Gp), _lib.pi(Gi), None, None, ctypes.byref(h)))
        snz = ctypes.c_int64()
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(snz), None, None))
        Sp = np.empty(n + 1, dtype=np.int64)
        Si = np.empty(max(snz.value, 1), dtype=np.int64)
        raise_for(lib().kvx_atda_pattern(h, ctypes.byref(snz), _lib.pi(Sp), _lib.pi(Si)))
        P = (h, Sp, Si[:snz.value].copy())
        if len(_plans) > 16:                         # evict everything, releasing the device side of each plan
            for old in _plans.values():
                lib().kvx_atda_free(old[0])
            _plans.clear()
        _plans[key] = P
    return P


def gemm(A, B, C, transA="N", transB="N", alpha=1.0, beta=0.0, partial=False):
    """C := diag(d) * B on C's (= B's) pattern -- the only sparse gemm on the hot path
    (misc.py:1418-1419: base.gemm(spdiag(W['di']), G, Gs, partial=True) -> sparse.c:1260-1283)."""
    if not partial or transA != "N" or transB != "N" or alpha != 1.0 or beta != 0.0:
        raise NotImplementedError("only gemm(spdiag(d), B, C, partial=True) is on the hot path")
    am, an, acp, ari, av = _as_ccs(A)
    bm, bn, bcp, bri, bv = _as_ccs(B)
    if am != an or an != bm or not np.array_equal(ari, np.arange(am)) or not np.array_equal(acp, np.arange(am + 1)):
        raise NotImplementedError("A must be a full sparse diagonal")
    if C.size != (bm, bn) or not np.array_equal(C.colptr, bcp) or not np.array_equal(C.rowind, bri):
        raise TypeError("C must have the pattern of B")
    _lib.require_device()
    # row scale as the 'T' product of a diagonal: do it with the NT scale kernel on a gathered copy
    d_v = DeviceBuffer.from_array(bv)
    d_w = DeviceBuffer.from_array(np.ascontiguousarray(av[bri]))
    raise_for(lib().kvx_nt_scale_dev(bv.size, 1, max(bv.size, 1), d_v.ptr, d_w.ptr))
    raise_for(lib().kvx_dev_sync())
    C.values[:] = d_v.download(np.float64, bv.size)


def syrk(A, C, uplo="L", trans="N", alpha=1.0, beta=0.0, partial=False):
    """C := alpha * A' * A + beta * C (lower triangle) for sparse A with trans='T'
    (misc.py:1422,1451 -> sparse.c:2173-2256).  partial=True keeps C's pattern; otherwise C's
    pattern is replaced by that of tril(A'A) (base.c:960-963)."""
    if trans != "T" or uplo != "L":
        raise NotImplementedError("only syrk(A, C, trans='T') is on the hot path")
    m, n, cp, ri, v = _as_ccs(A)
    if n == 0 or C.size[0] == 0:
        return                                              # base.c:914 early return on empty operands
    _lib.require_device()
    h, Sp, Si = _atda_plan(m, n, cp, ri)
    Sx = np.empty(max(Si.size, 1))
    w = np.full(max(m, 1), float(alpha))
    raise_for(lib().kvx_atda_assemble(h, _lib.pd(np.ascontiguousarray(v)), _lib.pd(w), None, _lib.pd(Sx)))
    Sx = Sx[:Si.size]
    if partial:
        # fixed pattern of C: pick the entries of A'A that C stores (sparse.c:2176-2198)
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Sp))
        key_full = cols * n + Si
        key_c = np.repeat(np.arange(n, dtype=np.int64), np.diff(C.colptr)) * n + C.rowind
        pos = np.searchsorted(key_full, key_c)
        pos = np.minimum(pos, max(key_full.size - 1, 0))
        hit = key_full[pos] == key_c if key_full.size else np.zeros(key_c.size, bool)
        new = np.where(hit, Sx[pos] if Sx.size else 0.0, 0.0)
        C.values[:] = new + (beta * C.values if beta != 0.0 else 0.0)
    else:
        if beta != 0.0:
            R = spmatrix.from_ccs(n, n, Sp, Si, Sx)
            tmp = spmatrix.from_ccs(C.size[0], C.size[1], C.colptr, C.rowind, beta * C.values)
            R += tmp
            C.colptr, C.rowind, C.values = R.colptr, R.rowind, R.values
        else:
            C.colptr, C.rowind, C.values = Sp.copy(), Si.copy(), Sx.copy()
        C.size = (n, n)


def gemv(A, x, y, trans="N", alpha=1.0, beta=0.0, m=None, n=None, incx=1, incy=1, offsetA=0, offsetx=0, offsety=0):
    """y := alpha*op(A)*x + beta*y (base.c:744-851 -> sparse.c:1073-1104 for sparse A), with the reference's sub-block
    selection: the product uses the m x n block of A whose top-left entry is A[offsetA % nrows, offsetA // nrows]
    (m, n default to A's size), x / y are read and written with strides incx / incy from offsetx / offsety
    (a negative stride walks the vector backwards, as in BLAS)."""
    xb, _ = _dense_buffer(x)
    yb, _ = _dense_buffer(y)
    if trans not in ("N", "T"):
        raise ValueError("possible values of trans are: 'N', 'T'")
    if incx == 0:
        raise ValueError("incx must be a nonzero integer")
    if incy == 0:
        raise ValueError("incy must be a nonzero integer")
    if isinstance(A, matrix) or isinstance(A, np.ndarray):
        # dense operand: every entry stored, same device kernel (no host arithmetic on the product path)
        Ad = np.asarray(A.a if isinstance(A, matrix) else A, dtype=np.float64)
        Ad = Ad.reshape(Ad.shape[0], -1)
        dm, dn = Ad.shape
        A = spmatrix.from_ccs(dm, dn, np.arange(dn + 1, dtype=np.int64) * dm, np.tile(np.arange(dm, dtype=np.int64), dn),
                              np.ascontiguousarray(Ad.T).reshape(-1))
    am, an, cp, ri, v = _as_ccs(A)
    m = am if m is None or m < 0 else int(m)
    n = an if n is None or n < 0 else int(n)
    if offsetA < 0:
        raise ValueError("offsetA must be a nonnegative integer")
    if n > 0 and m > 0 and offsetA + (n - 1) * max(1, am) + m > am * an:
        raise TypeError("length of A is too small")
    lx, ly = (n, m) if trans == "N" else (m, n)
    if offsetx < 0 or offsety < 0:
        raise ValueError("offsetx and offsety must be nonnegative integers")
    if lx > 0 and offsetx + (lx - 1) * abs(incx) + 1 > xb.size:
        raise TypeError("length of x is too small")
    if ly > 0 and offsety + (ly - 1) * abs(incy) + 1 > yb.size:
        raise TypeError("length of y is too small")
    if ly == 0:
        return
    ys = yb[offsety:offsety + (ly - 1) * abs(incy) + 1:abs(incy)]
    if incy < 0:
        # the reference scales y through BLAS dscal (sparse.c:1079), which returns at once for a non-positive stride:
        # with incy < 0 beta is NOT applied.  Kept, so that the results agree (golden G13, case 5).
        ys = ys[::-1]
        beta = 1.0
    if (m, n) != (am, an) or offsetA:
        # the selected block as a CCS matrix of its own (index work only; the product runs on the device)
        oi, oj = (offsetA % am, offsetA // am) if am else (0, 0)
        lo, hi = int(cp[oj]), int(cp[oj + n])
        rsel, vsel = ri[lo:hi], v[lo:hi]
        keep = (rsel >= oi) & (rsel < oi + m)
        col = np.repeat(np.arange(n, dtype=np.int64), np.diff(cp[oj:oj + n + 1]))[keep]
        cp = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(col, minlength=n), out=cp[1:])
        ri, v = np.ascontiguousarray(rsel[keep] - oi), np.ascontiguousarray(vsel[keep])
        am, an = m, n
    if lx == 0 or v.size == 0 or m == 0:
        # sparse.c:1079: y is scaled by beta and nothing is added (beta = 0 clears y, NaNs included)
        ys[:] = 0.0 if beta == 0.0 else beta * ys
        return
    xs = xb[offsetx:offsetx + (lx - 1) * abs(incx) + 1:abs(incx)]
    if incx < 0:
        xs = xs[::-1]
    _lib.require_device()
    d_cp, d_ri, d_v = DeviceBuffer.from_array(cp), DeviceBuffer.from_array(ri), DeviceBuffer.from_array(v)
    d_x = DeviceBuffer.from_array(np.ascontiguousarray(xs))
    d_y = DeviceBuffer.from_array(np.zeros(ly) if beta == 0.0 else np.ascontiguousarray(ys))
    raise_for(lib().kvx_spmv_dev(ord(trans), am, an, d_cp.ptr, d_ri.ptr, d_v.ptr, float(alpha), d_x.ptr, float(beta), d_y.ptr))
    raise_for(lib().kvx_dev_sync())
    ys[:] = d_y.download(np.float64, ly)
