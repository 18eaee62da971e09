"""Drop-in for `kvxopt.cholmod` on MI355X: same functions, argument meaning and error behaviour as
the reference's src/C/cholmod.c (method table cholmod.c:988-1005), backed by the HIP supernodal
Cholesky of libkvxhip.so.  No CPU fallback: numeric calls raise RuntimeError without a GPU.

    symbolic(A, p=None, uplo='L') -> F            cholmod.c:244-291
    numeric(A, F)                                  cholmod.c:322-398
    solve(F, B, sys=0, nrhs=-1, ldB=0, offsetB=0)  cholmod.c:429-499
    spsolve(F, B, sys=0) -> X                      cholmod.c:524-587
    linsolve(A, B, p=None, uplo='L', nrhs=-1, ldB=0, offsetB=0)   cholmod.c:618-753
    splinsolve(A, B, p=None, uplo='L') -> X        cholmod.c:774-881
    diag(F) -> d                                   cholmod.c:900-945
    getfactor(F) -> L                              cholmod.c:948-985
    options                                        dict, validated on every call (cholmod.c:87-129); 'nmethods' as cholmod.c:65-76

Differences kept on purpose: `numeric` raises ArithmeticError(minor) on a non-positive-
definite matrix as DOCUMENTED (cholmod.c:308-310); the reference build tests a stale status there
(quirk Q1) and only fails at the next solve -- `solve` here raises as well.
options['supernodal'] (spsolvers.rst:731-736): 2 (default) gives P A P' = L L'; 0 gives P A P' = L D L' and 1 chooses by
CHOLMOD's flops / nnz(L) >= 40 rule.  One set of HIP kernels computes the LL' factor Lc; the LDL' factor is that result seen
as L = Lc diag(Lc)^-1, D = diag(Lc)^2 -- `solve`/`spsolve` with sys = 2..6, `getfactor` (D on the diagonal) and `diag`
(refused, cholmod.c:919-922) follow the reference's behaviour for such a factor.
"""
import numpy as np

from . import base
from .base import matrix, spmatrix
from .chol import Factor

options = {}


def _check_options():
    """cholmod.c:87-129: unknown keys or wrongly typed values -> ValueError, on every entry."""
    opts = {}
    for k, v in options.items():
        if not isinstance(k, str):
            continue
        if k == "supernodal" and isinstance(v, int) and not isinstance(v, bool):
            opts["supernodal"] = v
        elif k == "print" and isinstance(v, int) and not isinstance(v, bool):
            pass
        elif k == "nmethods" and isinstance(v, int) and not isinstance(v, bool):
            opts["nmethods"] = v
        elif k == "postorder" and isinstance(v, bool):
            opts["postorder"] = int(v)
        elif k == "dbound" and isinstance(v, float):
            opts["dbound"] = v
        else:
            raise ValueError("invalid value for CHOLMOD parameter: %-.20s" % k)
    if opts.get("supernodal", 2) not in (0, 1, 2):
        raise ValueError("invalid value for CHOLMOD parameter: supernodal")
    return opts


class _F:
    """Opaque factor (the reference returns a PyCapsule named 'CHOLMOD SYM D FACTOR L'/'U')."""

    def __init__(self, fac, uplo, pattern, keep, tri):
        self.fac = fac
        self.uplo = uplo
        self.pattern = pattern          # (colptr, rowind) of the matrix symbolic() was given
        self.keep = keep                # mask of its entries inside the `uplo` triangle (None: all of them)
        self.tri = tri                  # (colptr, rowind) of that triangle: what was analysed (cholmod.c:132-181 `pack`)
        self.name = "CHOLMOD SYM D FACTOR " + uplo


def _triangle(n, cp, ri, uplo):
    """The `uplo` triangle of a CCS pattern as the reference's pack() reads it (cholmod.c:137-157): entries in the
    other triangle are ignored.  Returns (keep mask or None, colptr, rowind) of the triangle."""
    col = np.repeat(np.arange(n, dtype=np.int64), np.diff(cp))
    keep = ri >= col if uplo == "L" else ri <= col
    if keep.all():
        return None, cp, ri
    tcp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(col[keep], minlength=n), out=tcp[1:])
    return keep, tcp, ri[keep]


def _sp(A, what="A"):
    if not (isinstance(A, spmatrix) or hasattr(A, "CCS")):
        raise TypeError("%s is not a square sparse matrix" % what)
    return base._as_ccs(A)


def _perm(p, n):
    if p is None:
        return None
    if isinstance(p, matrix):
        if p.typecode != "i":
            raise TypeError("p must be a matrix with typecode 'i'")
        q = p._a
    elif hasattr(p, "typecode"):
        if p.typecode != "i":
            raise TypeError("p must be a matrix with typecode 'i'")
        q = np.array(list(p), dtype=np.int64)
    else:
        q = np.asarray(p)
        if q.dtype.kind not in "iu":
            raise TypeError("p must be a matrix with typecode 'i'")
    q = np.asarray(q, dtype=np.int64).reshape(-1)
    if q.size != n:
        raise TypeError("length of p is too small")
    return q


def symbolic(A, p=None, uplo="L"):
    opts = _check_options()
    m, n, cp, ri, v = _sp(A)
    if m != n:
        raise TypeError("A is not a square sparse matrix")
    q = _perm(p, n)
    if uplo not in ("L", "U"):
        raise ValueError("possible values of uplo are: 'L', 'U'")
    keep, tcp, tri = _triangle(n, cp, ri, uplo)
    fopts = {k: opts[k] for k in ("postorder", "dbound", "supernodal") if k in opts}
    # options['nmethods'] (cholmod.c:65-76): 1 = the given ordering and nothing else (no p: the natural order); 0 (default) and
    # 2 = a given p competes with the library's own orderings (nested dissection / minimum degree), the least fill wins
    nmethods = opts.get("nmethods", 0)
    if nmethods == 1:
        if q is None:
            fopts["ordering"] = 1
    elif q is not None:
        fopts["compare_given"] = 1
    try:
        fac = Factor(n, tcp, tri, uplo, q, fopts)
    except ValueError as e:
        if "permutation" in str(e):
            raise ValueError("p is not a valid permutation")
        raise
    return _F(fac, uplo, (cp.copy(), ri.copy()), keep, (tcp.copy(), tri.copy()))


def numeric(A, F):
    _check_options()
    m, n, cp, ri, v = _sp(A)
    if m != n:
        raise TypeError("A is not a sparse matrix")
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    if n != F.fac.n:
        raise ValueError("factorization failed")
    # only the `uplo` triangle counts (cholmod.c:137-157): same full pattern as analysed -> reuse its mask; otherwise the
    # triangle of this A must have the analysed triangle's pattern exactly (same nnz with other positions is an error,
    # a matrix that differs only in the ignored triangle is accepted)
    if ri.size == F.pattern[1].size and np.array_equal(cp, F.pattern[0]) and np.array_equal(ri, F.pattern[1]):
        vt = v if F.keep is None else v[F.keep]
    else:
        keep, tcp, tri = _triangle(n, cp, ri, F.uplo)
        if tri.size != F.tri[1].size or not np.array_equal(tcp, F.tri[0]) or not np.array_equal(tri, F.tri[1]):
            raise ValueError("factorization failed: A does not have the sparsity pattern of the symbolic factorization")
        vt = v if keep is None else v[keep]
    F.fac.factorize(vt)                                  # ArithmeticError(minor) if not positive definite


def _solve_args(F, B, sys, nrhs, ldB, offsetB):
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    if not F.fac.info()["is_numeric"] and F.fac.info()["minor"] == F.fac.n:
        raise ValueError("called with symbolic factor")
    if sys < 0 or sys > 8:
        raise ValueError("invalid value for sys")
    buf, size = base._dense_buffer(B)
    n = F.fac.n
    if nrhs < 0:
        nrhs = size[1]
    return buf, size, n, nrhs


def solve(F, B, sys=0, nrhs=-1, ldB=0, offsetB=0):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    inf = F.fac.info()
    if not inf["is_numeric"] and inf["minor"] >= F.fac.n:
        raise ValueError("called with symbolic factor")
    if inf["minor"] < F.fac.n:
        raise ArithmeticError("singular matrix")
    if sys < 0 or sys > 8:
        raise ValueError("invalid value for sys")
    buf, size = base._dense_buffer(B)
    n = F.fac.n
    if nrhs < 0:
        nrhs = size[1]
    if n == 0 or nrhs == 0:
        return
    if ldB == 0:
        ldB = max(1, size[0])
    if ldB < max(1, n):
        raise ValueError("illegal value of ldB")
    if offsetB < 0:
        raise ValueError("offsetB must be a nonnegative integer")
    if offsetB + (nrhs - 1) * ldB + n > buf.size:
        raise TypeError("length of B is too small")
    F.fac.solve(buf, sys=sys, nrhs=nrhs, ldB=ldB, offset=offsetB)


def spsolve(F, B, sys=0):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    inf = F.fac.info()
    if not inf["is_numeric"] and inf["minor"] >= F.fac.n:
        raise ValueError("called with symbolic factor")
    if inf["minor"] < F.fac.n:
        raise ArithmeticError("singular matrix")
    if sys < 0 or sys > 8:
        raise ValueError("invalid value for sys")
    m, ncol, cp, ri, v = _sp(B, "B")
    if m != F.fac.n:
        raise ValueError("incompatible dimensions for B")
    if F.fac.n == 0 or ncol == 0:
        return spmatrix([], [], [], (m, ncol))
    Xp, Xi, Xx = F.fac.spsolve(ncol, cp, ri, v, sys)
    return spmatrix.from_ccs(m, ncol, Xp, Xi, Xx)


def linsolve(A, B, p=None, uplo="L", nrhs=-1, ldB=0, offsetB=0):
    _check_options()
    m, n, cp, ri, v = _sp(A)
    if m != n:
        raise TypeError("A is not a sparse matrix")
    buf, size = base._dense_buffer(B)
    if nrhs < 0:
        nrhs = size[1]
    if n == 0 or nrhs == 0:
        return
    F = symbolic(A, p, uplo)
    numeric(A, F)
    solve(F, B, 0, nrhs, ldB, offsetB)


def splinsolve(A, B, p=None, uplo="L"):
    _check_options()
    m, n, cp, ri, v = _sp(A)
    if m != n:
        raise TypeError("A is not a square sparse matrix")
    bm = B.size[0]
    if bm != n:
        raise ValueError("incompatible dimensions for B")
    F = symbolic(A, p, uplo)
    numeric(A, F)
    return spsolve(F, B, 0)


def diag(F):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    inf = F.fac.info()
    if not inf["is_numeric"] or inf["minor"] < F.fac.n or not inf["is_ll"]:
        raise ValueError("F must be a nonsingular supernodal Cholesky factor")
    return matrix(F.fac.diag(), (F.fac.n, 1))


def getfactor(F):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    if not F.fac.info()["is_numeric"]:
        raise ValueError("F must be a numeric Cholesky factor")
    Lp, Li, Lx = F.fac.get_factor()
    return spmatrix.from_ccs(F.fac.n, F.fac.n, Lp, Li, Lx)
