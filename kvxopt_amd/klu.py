"""Drop-in for `kvxopt.klu` on MI355X: same functions, argument meaning and error behaviour as the reference's
src/C/klu.c (method table klu.c:831-840), backed by the HIP multifrontal LU of libkvxhip.so (kvx_lu_*).
No CPU fallback: numeric calls raise RuntimeError without a GPU.

    linsolve(A, B, trans='N', nrhs=-1, ldB=0, offsetB=0)        klu.c:94-230
    symbolic(A) -> Fs                                            klu.c:242-291
    numeric(A, Fs[, Fn]) -> Fn                                   klu.c:310-379 (refactorisation: spsolvers.rst:377-388)
    solve(A, Fs, Fn, B, trans='N', nrhs=-1, ldB=0, offsetB=0)    klu.c:593-690
    get_numeric(A, Fs, Fn) -> L, U, P, Q, R, F, r                klu.c:392-566   (R P A Q = L U + F)
    get_det(A, Fs, Fn) -> float                                  klu.c:707-828

Real ('d') matrices only; complex ('z') raises TypeError (out of scope).  The factorisation has KLU's form
(block triangular permutation, L U of the diagonal blocks, off-diagonal part F, row scaling R) but not its algorithm:
static multifrontal fronts with in-front threshold pivoting -- the identities the reference's tests check
(tests/test_sparse_solvers.py:214-323) hold all the same.
"""
import numpy as np

from . import base
from .base import matrix, spmatrix
from .lu import LuSymbolic, LuNumeric


class _Fs:
    """Opaque symbolic factor (the reference returns a PyCapsule named 'KLU SYM D FACTOR', klu.c:36)."""
    name = "KLU SYM D FACTOR"

    def __init__(self, sym, pattern):
        self.sym = sym
        self.pattern = pattern


class _Fn:
    """Opaque numeric factor ('KLU NUM D FACTOR', klu.c:38)."""
    name = "KLU NUM D FACTOR"

    def __init__(self, num):
        self.num = num


def _sp(A, square=True, msg="A must be a square sparse matrix"):
    if not (isinstance(A, spmatrix) or hasattr(A, "CCS")):
        raise TypeError(msg)
    if getattr(A, "typecode", "d") != "d":
        raise TypeError("kvxopt_amd.klu implements real ('d') matrices only")
    m, n, cp, ri, v = base._as_ccs(A)
    if square and m != n:
        raise TypeError(msg)
    return n, cp, ri, v


def _same_pattern(Fs, cp, ri):
    return cp.size == Fs.pattern[0].size and ri.size == Fs.pattern[1].size and \
        np.array_equal(cp, Fs.pattern[0]) and np.array_equal(ri, Fs.pattern[1])


def symbolic(A):
    n, cp, ri, v = _sp(A)
    if n == 0:
        raise ValueError("A must have at least one row and column")
    return _Fs(LuSymbolic(n, cp, ri, v), (cp.copy(), ri.copy()))


def numeric(A, Fs, Fn=None):
    n, cp, ri, v = _sp(A, square=False, msg="A must be a sparse matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("Fs is not the KLU symbolic factor of a 'd' matrix")
    if not _same_pattern(Fs, cp, ri):
        raise ValueError("KLU ERROR -3")                     # KLU_INVALID: not the analysed pattern
    if Fn is not None:
        if not isinstance(Fn, _Fn) or Fn.num.sym is not Fs.sym:
            raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
        Fn.num.refactor(v)                                   # ArithmeticError("singular matrix") as klu.c:370-371
        return Fn
    return _Fn(LuNumeric(Fs.sym, v))


def _rhs_args(n, B, trans, nrhs, ldB, offsetB):
    if not (isinstance(B, (matrix, np.ndarray)) or hasattr(B, "typecode")):
        raise TypeError("B must a dense matrix of the same numeric type as A")
    buf, size = base._dense_buffer(B)
    if nrhs < 0:
        nrhs = size[1]
    if n == 0 or nrhs == 0:
        return buf, 0, ldB
    if ldB == 0:
        ldB = max(1, size[0])
    if ldB < max(1, n):
        raise ValueError("illegal value of ldB")
    if offsetB < 0:
        raise ValueError("offsetB must be a nonnegative integer")
    if offsetB + (nrhs - 1) * ldB + n > buf.size:
        raise TypeError("length of B is too small")
    if trans not in ("N", "T", "C"):
        raise ValueError("possible values of trans are: 'N', 'T', 'C'")
    return buf, nrhs, ldB


def solve(A, Fs, F, B, trans="N", nrhs=-1, ldB=0, offsetB=0):
    n, cp, ri, v = _sp(A, msg="A must a square sparse matrix")
    if not isinstance(F, _Fn):
        raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("F is not the KLU symbolic factor of a 'd' matrix")
    buf, nrhs, ldB = _rhs_args(n, B, trans, nrhs, ldB, offsetB)
    if nrhs == 0:
        return
    F.num.solve(buf, trans="N" if trans == "N" else "T", nrhs=nrhs, ldB=ldB, offset=offsetB)


def linsolve(A, B, trans="N", nrhs=-1, ldB=0, offsetB=0):
    n, cp, ri, v = _sp(A)
    buf, nrhs_, ldB_ = _rhs_args(n, B, trans, nrhs, ldB, offsetB)
    if nrhs_ == 0:
        return 0 if n == 0 or nrhs == 0 else None
    Fs = symbolic(A)
    Fn = numeric(A, Fs)
    solve(A, Fs, Fn, B, trans, nrhs, ldB, offsetB)


def get_numeric(A, Fs, Fn):
    n, cp, ri, v = _sp(A, square=False, msg="A must be a sparse matrix")
    if not isinstance(Fn, _Fn):
        raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("F is not the KLU symbolic factor of a 'd' matrix")
    e = Fn.num.extract()
    L = spmatrix.from_ccs(n, n, *e["L"])
    U = spmatrix.from_ccs(n, n, *e["U"])
    F = spmatrix.from_ccs(n, n, *e["F"])
    ar = np.arange(n, dtype=np.int64)
    # klu.c:503-528: R = diag(1 / Rs) so that R*P*A*Q (not R\P*A*Q) equals L*U + F; P(i, Pt[i]) = 1; Q(Qt[i], i) = 1
    R = spmatrix(1.0 / e["Rs"], ar, ar, (n, n))
    P = spmatrix(np.ones(n), ar, e["P"], (n, n))
    Q = spmatrix(np.ones(n), e["Q"], ar, (n, n))
    return L, U, P, Q, R, F, [int(x) for x in e["r"]]


def get_det(A, Fs, Fn):
    _sp(A, square=False, msg="A must be a sparse matrix")
    if not isinstance(Fn, _Fn):
        raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("F is not the KLU symbolic factor of a 'd' matrix")
    return Fn.num.det()
