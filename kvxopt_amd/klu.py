"""Drop-in for `kvxopt.klu` on MI355X: same functions, argument meaning and error behaviour as the reference's
src/C/klu.c (method table klu.c:831-840), backed by the HIP multifrontal LU of libkvxhip.so (kvx_lu_*).
No CPU fallback: numeric calls raise RuntimeError without a GPU.

    linsolve(A, B, trans='N', nrhs=-1, ldB=0, offsetB=0)        klu.c:94-230
    symbolic(A) -> Fs                                            klu.c:242-291
    numeric(A, Fs[, Fn]) -> Fn                                   klu.c:310-379 (refactorisation: spsolvers.rst:377-388)
    solve(A, Fs, Fn, B, trans='N', nrhs=-1, ldB=0, offsetB=0)    klu.c:593-690
    get_numeric(A, Fs, Fn) -> L, U, P, Q, R, F, r                klu.c:392-566   (R P A Q = L U + F)
    get_det(A, Fs, Fn) -> float                                  klu.c:707-828

Complex ('z') matrices (the reference's tests run every case with A + A*1j, tests/test_sparse_solvers.py:86-95) are SOLVED --
`linsolve`, `symbolic`, `numeric`, `solve` with trans 'N', 'T', 'C' -- through the real 2n x 2n embedding [[Re A, -Im A],
[Im A, Re A]] factored by the same real kernels (A^H is the transpose of the embedding; A^T x = b is conj(A^-H conj(b)));
`get_numeric` / `get_det` of a complex factor are not available (the factors of the embedding are not embeddings of complex
factors) and raise NotImplementedError.  The factorisation has KLU's form
(block triangular permutation, L U of the diagonal blocks, off-diagonal part F, row scaling R) but not its algorithm:
static multifrontal fronts with in-front threshold pivoting -- the identities the reference's tests check
(tests/test_sparse_solvers.py:214-323) hold all the same.
"""
import collections
import os

import numpy as np

from . import _lib, base
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
    if getattr(A, "typecode", "d") not in ("d", "z"):
        raise TypeError(msg)
    m, n, cp, ri, v = base._as_ccs(A)
    if square and m != n:
        raise TypeError(msg)
    return n, cp, ri, v


def _embed(n, cp, ri, v):
    """CCS of the real embedding [[Re A, -Im A], [Im A, Re A]] of a complex n x n CCS matrix (every stored entry keeps its
    place in all four blocks, zero imaginary parts included: the pattern depends on A's pattern only)."""
    nnz = ri.size
    cnt = np.diff(cp)
    cp2 = np.concatenate([2 * cp[:-1], 2 * nnz + 2 * cp]).astype(np.int64)
    ri2 = np.empty(4 * nnz, dtype=np.int64)
    v2 = np.empty(4 * nnz)
    pos = np.arange(nnz, dtype=np.int64) - np.repeat(cp[:-1], cnt)          # position inside its column
    base0 = 2 * np.repeat(cp[:-1], cnt)                                     # start of column j of the embedding
    top, bot = base0 + pos, base0 + np.repeat(cnt, cnt) + pos
    ri2[top], ri2[bot] = ri, ri + n
    v2[top], v2[bot] = v.real, v.imag
    top2, bot2 = 2 * nnz + top, 2 * nnz + bot                               # column n + j
    ri2[top2], ri2[bot2] = ri, ri + n
    v2[top2], v2[bot2] = -v.imag, v.real
    return 2 * n, cp2, ri2, v2


def _same_pattern(Fs, cp, ri):
    return cp.size == Fs.pattern[0].size and ri.size == Fs.pattern[1].size and \
        np.array_equal(cp, Fs.pattern[0]) and np.array_equal(ri, Fs.pattern[1])


def symbolic(A):
    n, cp, ri, v = _sp(A)
    if n == 0:
        raise ValueError("A must have at least one row and column")
    if v.dtype.kind == "c":
        Fs = _Fs(LuSymbolic(*_embed(n, cp, ri, v)), (cp.copy(), ri.copy()))
        Fs.name, Fs.complex = "KLU SYM Z FACTOR", True
        return Fs
    return _Fs(LuSymbolic(n, cp, ri, v), (cp.copy(), ri.copy()))


def numeric(A, Fs, Fn=None):
    n, cp, ri, v = _sp(A, square=False, msg="A must be a sparse matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("Fs is not the KLU symbolic factor of a 'd' matrix")
    if not _same_pattern(Fs, cp, ri):
        raise ValueError("KLU ERROR -3")                     # KLU_INVALID: not the analysed pattern
    if (v.dtype.kind == "c") != bool(getattr(Fs, "complex", False)):
        raise TypeError("Fs is not the KLU symbolic factor of a '%s' matrix" % ("z" if v.dtype.kind == "c" else "d"))
    if v.dtype.kind == "c":
        v = _embed(n, cp, ri, v)[3]
    if Fn is not None:
        if not isinstance(Fn, _Fn) or Fn.num.sym is not Fs.sym:
            raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
        Fn.num.refactor(v)                                   # ArithmeticError("singular matrix") as klu.c:370-371
        return Fn
    return _Fn(LuNumeric(Fs.sym, v))


def _rhs_args(n, B, trans, nrhs, ldB, offsetB, tc="d"):
    if not (isinstance(B, (matrix, np.ndarray)) or hasattr(B, "typecode")):
        raise TypeError("B must a dense matrix of the same numeric type as A")
    try:
        buf, size = base._dense_buffer(B, tc)
    except TypeError:
        raise TypeError("B must a dense matrix of the same numeric type as A")
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
    cplx = v.dtype.kind == "c"
    if cplx != bool(getattr(Fs, "complex", False)):
        raise TypeError("F is not the KLU numeric factor of a '%s' matrix" % ("z" if cplx else "d"))
    buf, nrhs, ldB = _rhs_args(n, B, trans, nrhs, ldB, offsetB, "z" if cplx else "d")
    if nrhs == 0:
        return
    if not cplx:
        F.num.solve(buf, trans="N" if trans == "N" else "T", nrhs=nrhs, ldB=ldB, offset=offsetB)
        return
    # complex: [Re; Im] through the embedding.  'C': A^H = transpose of the embedding; 'T': A^T x = b  <=>  A^H conj(x) = conj(b)
    cols = [buf[offsetB + j * ldB: offsetB + j * ldB + n] for j in range(nrhs)]
    R = np.empty((2 * n, nrhs), order="F")
    for j, c in enumerate(cols):
        R[:n, j], R[n:, j] = c.real, (-c.imag if trans == "T" else c.imag)
    F.num.solve(R.reshape(-1, order="F"), trans="N" if trans == "N" else "T", nrhs=nrhs, ldB=2 * n, offset=0)
    for j, c in enumerate(cols):
        c[:] = R[:n, j] + 1j * (-R[n:, j] if trans == "T" else R[n:, j])


# linsolve re-analyses and re-factors at every call in the reference (klu.c:142-198).  Here the symbolic and numeric factors of the
# last few sparsity patterns are kept: a repeated call with a known pattern is a numeric REfactorisation (same pivot sequence,
# full factorisation as fall-back, doc/source/spsolvers.rst:377-388) plus a solve -- the host analysis (matching, block
# triangular form, ordering: 4-5 ms on ACTIVSg2000) is what made a one-shot call slower than a host SuperLU.
_LINSOLVE_CACHE = collections.OrderedDict()
_LINSOLVE_CACHE_MAX = int(os.environ.get("KVX_LINSOLVE_CACHE", "8"))      # 0 turns the cache off (every kept factor holds device memory)


def clear_cache():
    """Release the factors kept for linsolve (device memory)."""
    _LINSOLVE_CACHE.clear()


_lib.register_cache(clear_cache)


def linsolve(A, B, trans="N", nrhs=-1, ldB=0, offsetB=0):
    n, cp, ri, v = _sp(A)
    buf, nrhs_, ldB_ = _rhs_args(n, B, trans, nrhs, ldB, offsetB, "z" if v.dtype.kind == "c" else "d")
    if nrhs_ == 0:
        return 0 if n == 0 or nrhs == 0 else None
    key = (n, v.dtype.kind, _lib.pattern_digest(cp, ri), _lib.current_device())     # the factors live on ONE device
    hit = _LINSOLVE_CACHE.pop(key, None) if _LINSOLVE_CACHE_MAX > 0 else None
    if hit is not None and not _same_pattern(hit[0], cp, ri):  # (a digest is not the pattern)
        hit = None
    if hit is not None:
        Fs, Fn = hit
        try:
            if v.dtype.kind == "c":
                Fn = numeric(A, Fs, Fn)
            else:
                Fn.num.refactor(v)                        # (numeric(A, Fs, Fn) without its second comparison of the pattern)
        except ArithmeticError:
            hit = None                                    # (singular with these values: start over below and report from there)
    if hit is None:
        Fs = symbolic(A)
        Fn = numeric(A, Fs)
    if _LINSOLVE_CACHE_MAX > 0:
        _LINSOLVE_CACHE[key] = (Fs, Fn)
    while len(_LINSOLVE_CACHE) > _LINSOLVE_CACHE_MAX:
        _LINSOLVE_CACHE.popitem(last=False)
    solve(A, Fs, Fn, B, trans, nrhs, ldB, offsetB)


def get_numeric(A, Fs, Fn):
    n, cp, ri, v = _sp(A, square=False, msg="A must be a sparse matrix")
    if not isinstance(Fn, _Fn):
        raise TypeError("F is not the KLU numeric factor of a 'd' matrix")
    if not isinstance(Fs, _Fs):
        raise TypeError("F is not the KLU symbolic factor of a 'd' matrix")
    if getattr(Fs, "complex", False):
        raise NotImplementedError("get_numeric of a complex factor: complex systems are solved through their real embedding, "
                                  "whose factors are not the complex L, U")
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
    if getattr(Fs, "complex", False):
        raise NotImplementedError("get_det of a complex factor (the embedding only gives |det A|^2)")
    return Fn.num.det()
