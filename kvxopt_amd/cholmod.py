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
Complex Hermitian ('z') matrices (cholmod.c:144,153,463) are SOLVED -- symbolic / numeric / solve and spsolve with sys = 0,
linsolve, splinsolve, and since round 4 every system code sys = 0..8, `diag` and `getfactor` -- through the real symmetric
2n x 2n embedding in interleaved numbering (every a_ij a 2 x 2 block [[Re, -Im], [Im, Re]]; positive definite exactly when A
is) on the same real kernels: eliminated pair by pair in a complex elimination order, the real factor IS the embedding of the
complex one (`_embed_hermitian`), and a complex vector in memory is its own embedding.  Only LDL' factors of complex matrices
(options['supernodal'] = 0) keep to sys = 0 / diag-free use.
"""
import collections
import os

import numpy as np

from . import _lib, base
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


def _embed_hermitian(n, cp, ri, v, uplo):
    """Lower-triangular CCS (colptr, rowind, values, source map) of the real embedding of the Hermitian matrix A = B + iC given
    by its `uplo` triangle, in INTERLEAVED numbering (real part of unknown j at 2j, imaginary part at 2j + 1 -- the memory
    layout of a complex vector): every a_ij becomes the 2 x 2 block [[Re, -Im], [Im, Re]].  A stored a_ij, i > j, lands in
    (2i, 2j), (2i+1, 2j+1) with Re a_ij and in (2i+1, 2j), (2j+1, 2i) -> lower (2i, 2j+1) with +-Im a_ij; the (real part of
    the) diagonal in (2i, 2i), (2i+1, 2i+1).  Built once per pattern: the returned gather turns a value array of A into the
    value array of the embedding.
    Why interleaved: eliminated pair by pair in a complex elimination order, the real Cholesky factor R of the embedding IS
    the embedding of the complex factor L -- block (i, j) of R is [[Re l_ij, -Im l_ij], [Im l_ij, Re l_ij]], the diagonal
    blocks are l_jj I (uniqueness of the Cholesky factor: that matrix is lower triangular with a positive diagonal and its
    product with its transpose is the embedding of L L^H).  So every system of cholmod.solve, diag and getfactor of a complex
    factor are read off the real one."""
    col = np.repeat(np.arange(n, dtype=np.int64), np.diff(cp))
    keep = ri >= col if uplo == "L" else ri <= col
    idx = np.nonzero(keep)[0]
    i, j = ri[idx], col[idx]
    flip = np.zeros(idx.size, dtype=bool)
    if uplo == "U":                                   # stored a_ij with i <= j: the lower entry is a_ji = conj(a_ij)
        i, j = j, i
        flip[:] = True
    off = i > j
    # entries: (row, col, source index, kind)  kind 0: Re, 1: +Im, 2: -Im (before the conjugation of an upper triangle)
    # lower-triangle positions: Re at (2i, 2j) and (2i+1, 2j+1); Im a_ij at (2i+1, 2j); -Im a_ij (block entry (2i, 2j+1))
    rows = np.concatenate([2 * i, 2 * i + 1, 2 * i[off] + 1, 2 * i[off]])
    cols = np.concatenate([2 * j, 2 * j + 1, 2 * j[off], 2 * j[off] + 1])
    src = np.concatenate([idx, idx, idx[off], idx[off]])
    kind = np.concatenate([np.zeros(idx.size, np.int8), np.zeros(idx.size, np.int8), np.ones(off.sum(), np.int8),
                           np.full(off.sum(), 2, np.int8)])
    conj = np.concatenate([flip, flip, flip[off], flip[off]])
    order = np.lexsort((rows, cols))
    rows, cols, src, kind, conj = rows[order], cols[order], src[order], kind[order], conj[order]
    cp2 = np.zeros(2 * n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=2 * n), out=cp2[1:])
    sign = np.where(kind == 2, -1.0, 1.0) * np.where(conj & (kind > 0), -1.0, 1.0)

    def values(vz):
        vz = np.asarray(vz, dtype=np.complex128)
        return np.where(kind == 0, vz.real[src], sign * vz.imag[src])
    return cp2, np.ascontiguousarray(rows), values


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
    fopts = {k: opts[k] for k in ("postorder", "dbound", "supernodal") if k in opts}
    # options['nmethods'] (cholmod.c:65-76): 1 = the given ordering and nothing else (no p: the natural order); 0 (default) and
    # 2 = a given p competes with the library's own orderings (nested dissection / minimum degree), the least fill wins
    nmethods = opts.get("nmethods", 0)
    if nmethods == 1:
        if q is None:
            fopts["ordering"] = 1
    elif q is not None:
        fopts["compare_given"] = 1
    if v.dtype.kind == "c":
        # the ordering is chosen on the n x n pattern of A (host analysis only) and the embedding eliminates the two unknowns of
        # a complex one side by side in that order: the real factor is then the embedding of the complex one (_embed_hermitian)
        keep, tcp, tri = _triangle(n, cp, ri, uplo)
        try:
            f0 = Factor(n, tcp, tri, uplo, q, fopts)
        except ValueError as e:
            if "permutation" in str(e):
                raise ValueError("p is not a valid permutation")
            raise
        p0 = f0.perm()
        del f0
        q2 = np.empty(2 * n, dtype=np.int64)
        q2[0::2], q2[1::2] = 2 * p0, 2 * p0 + 1
        ecp, eri, evals = _embed_hermitian(n, cp, ri, v, uplo)
        fo2 = {k: fopts[k] for k in ("dbound", "supernodal") if k in fopts}
        fo2["postorder"] = 0                                # (p0 is a postorder of the complex tree; the pairs stay together)
        fac = Factor(2 * n, ecp, eri, "L", q2, fo2)
        F = _F(fac, uplo, (cp.copy(), ri.copy()), None, (ecp, eri))
        F.name, F.complex, F.evals, F.n = "CHOLMOD SYM Z FACTOR " + uplo, True, evals, n
        # The library postorders whatever it is given (supernodes need contiguous subtrees): siblings may change places -- Re and
        # Im of one unknown are siblings of the real tree -- which permutes the real factor without changing an entry.  What the
        # complex reading needs is that the two stay next to each other; which of them comes first is kept per pair.
        P2 = fac.perm()
        F.paired = bool(np.all(P2[0::2] // 2 == P2[1::2] // 2))
        F.pair_swap = (P2[0::2] % 2) == 1                    # pair t has its imaginary part first
        F.pair_perm = P2[0::2] // 2                          # the complex permutation (sys = 7 / 8)
        F.real_perm = P2
        return F
    keep, tcp, tri = _triangle(n, cp, ri, uplo)
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
    if getattr(F, "complex", False) != (v.dtype.kind == "c"):
        raise TypeError("F is not the CHOLMOD factor of a '%s' matrix" % ("z" if v.dtype.kind == "c" else "d"))
    if getattr(F, "complex", False):
        if n != F.n or ri.size != F.pattern[1].size or not np.array_equal(cp, F.pattern[0]) or not np.array_equal(ri, F.pattern[1]):
            raise ValueError("factorization failed: A does not have the sparsity pattern of the symbolic factorization")
        F.fac.factorize(F.evals(v))
        return
    F.fac.factorize(_triangle_values(F, n, cp, ri, v))    # ArithmeticError(minor) if not positive definite


def _triangle_values(F, n, cp, ri, v, same_pattern=False):
    """The values of A's `uplo` triangle in the order of the analysed pattern (what numeric hands to the factorisation).
    same_pattern: the caller has established that (cp, ri) is the analysed pattern (linsolve found F under the pattern's digest:
    comparing config 2's 32 MB of indices again was 3.5 ms of an 8 ms call)."""
    if n != F.fac.n:
        raise ValueError("factorization failed")
    if same_pattern and ri.size == F.pattern[1].size:
        return v if F.keep is None else v[F.keep]
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
    return vt


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
    cplx = getattr(F, "complex", False)
    try:
        buf, size = base._dense_buffer(B, "z" if cplx else "d")
    except TypeError:
        raise TypeError("B must a dense matrix of the same numeric type as F")       # cholmod.c:461-465
    n = F.n if cplx else F.fac.n
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
    if cplx:
        # a complex vector in memory IS its interleaved embedding: the same system code on the real factor, in place
        # (L x = b <-> R e(x) = e(b), L^H <-> R', the permutation of the pairs <-> P; cholmod.c:463 z-typed solve)
        if sys != 0 and not getattr(F, "paired", False):
            raise NotImplementedError("this complex factor's ordering split a (Re, Im) pair: only A X = B (sys = 0) is available")
        fb = buf.view(np.float64)
        sw = F.pair_swap if sys != 0 and F.pair_swap.any() else None

        def swap_pairs():                                   # vectors in the factor's permuted order: pairs stored Im first
            for j in range(nrhs):
                c = fb[2 * (offsetB + j * ldB): 2 * (offsetB + j * ldB + n)].reshape(n, 2)
                c[sw] = c[sw][:, ::-1]
        if sw is not None and sys != 7:                     # (sys = 7 takes b in the caller's order)
            swap_pairs()
        F.fac.solve(fb, sys=sys, nrhs=nrhs, ldB=2 * ldB, offset=2 * offsetB)
        if sw is not None and sys != 8:                     # (sys = 8 returns x in the caller's order)
            swap_pairs()
        return
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
    if getattr(F, "complex", False):
        if m != F.n:
            raise ValueError("incompatible dimensions for B")
        # columns of B as dense complex vectors through the embedding; exact zeros are dropped as CHOLMOD's own spsolve does
        n = F.n
        D = np.zeros((n, ncol), dtype=np.complex128, order="F")
        D[ri, np.repeat(np.arange(ncol, dtype=np.int64), np.diff(cp))] = v
        solve(F, D, sys)
        I, J = np.nonzero(D.T != 0)
        return spmatrix(D[J, I], J, I, (n, ncol), "z")
    if m != F.fac.n:
        raise ValueError("incompatible dimensions for B")
    if F.fac.n == 0 or ncol == 0:
        return spmatrix([], [], [], (m, ncol))
    Xp, Xi, Xx = F.fac.spsolve(ncol, cp, ri, v, sys)
    return spmatrix.from_ccs(m, ncol, Xp, Xi, Xx)


# linsolve / splinsolve analyse the matrix at every call in the reference (cholmod.c:663, 811).  The symbolic factors of the last few
# (pattern, p, uplo, options) combinations are kept here: a repeated call is a numeric refactorisation + solve; the host analysis
# (0.2 s at n = 1e6) is 98 % of a one-shot call otherwise.
_SYMBOLIC_CACHE = collections.OrderedDict()
_SYMBOLIC_CACHE_MAX = int(os.environ.get("KVX_LINSOLVE_CACHE", "4"))      # 0 turns the cache off (every kept factor holds device memory)


def clear_cache():
    """Release the symbolic factors kept for linsolve / splinsolve (device memory, launch graphs)."""
    _SYMBOLIC_CACHE.clear()


_lib.register_cache(clear_cache)


def _cached_symbolic(A, p, uplo, n, cp, ri, v):
    q = _perm(p, n)
    # (a 128-bit digest of the index arrays, not their bytes: copying and hashing the 32 MB of config 2's pattern was 22 ms of
    #  every call on a known pattern -- the whole call is 6 ms with the digest)
    key = (n, uplo, v.dtype.kind, None if q is None else _lib.pattern_digest(q), _lib.pattern_digest(cp, ri),
           tuple(sorted((k, repr(val)) for k, val in options.items())), _lib.current_device())   # a factor lives on ONE device
    if _SYMBOLIC_CACHE_MAX <= 0:
        return symbolic(A, p, uplo)
    F = _SYMBOLIC_CACHE.pop(key, None)
    if F is None:
        F = _lib.retry_after_release(lambda: symbolic(A, p, uplo))
    _SYMBOLIC_CACHE[key] = F
    while len(_SYMBOLIC_CACHE) > _SYMBOLIC_CACHE_MAX:
        _SYMBOLIC_CACHE.popitem(last=False)
    return F


def linsolve(A, B, p=None, uplo="L", nrhs=-1, ldB=0, offsetB=0):
    _check_options()
    m, n, cp, ri, v = _sp(A)
    if m != n:
        raise TypeError("A is not a sparse matrix")
    buf, size = base._dense_buffer(B, "z" if v.dtype.kind == "c" else "d")
    if nrhs < 0:
        nrhs = size[1]
    if n == 0 or nrhs == 0:
        return
    F = _cached_symbolic(A, p, uplo, n, cp, ri, v)
    # numeric + solve as ONE call into the library (kvx_chol_factorize_solve: the forward sweep runs beside the factorisation of
    # the top of the tree) for the plain case -- a real LL' factor, a handful of right-hand sides, B checked as solve() checks it
    if v.dtype.kind != "c" and 0 < nrhs <= 16 and F.fac.info()["is_ll"]:
        ld = ldB if ldB else max(1, size[0])
        if ld >= max(1, n) and offsetB >= 0 and offsetB + (nrhs - 1) * ld + n <= buf.size:
            F.fac.factorize_solve(_triangle_values(F, n, cp, ri, v, same_pattern=True), buf, nrhs=nrhs, ldB=ld, offset=offsetB)
            return
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
    F = _cached_symbolic(A, p, uplo, n, cp, ri, v)
    numeric(A, F)
    return spsolve(F, B, 0)


def diag(F):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    inf = F.fac.info()
    if not inf["is_numeric"] or inf["minor"] < F.fac.n or not inf["is_ll"]:
        raise ValueError("F must be a nonsingular supernodal Cholesky factor")
    if getattr(F, "complex", False):
        # cholmod.c:900-945 on a 'z' factor: the diagonal of L (real and positive) as a 'z' matrix; the diagonal blocks of the
        # real factor are l_jj I
        if not getattr(F, "paired", False):
            raise NotImplementedError("this complex factor's ordering split a (Re, Im) pair: diag is not available")
        return matrix(F.fac.diag()[0::2].astype(np.complex128), (F.n, 1))
    return matrix(F.fac.diag(), (F.fac.n, 1))


def getfactor(F):
    _check_options()
    if not isinstance(F, _F):
        raise TypeError("F is not a CHOLMOD factor")
    if not F.fac.info()["is_numeric"]:
        raise ValueError("F must be a numeric Cholesky factor")
    if getattr(F, "complex", False):
        # cholmod.c:948-985 on a 'z' factor: column 2j of the real factor holds Re l_ij at row 2i and Im l_ij at row 2i + 1
        if not getattr(F, "paired", False) or not F.fac.info()["is_ll"]:
            raise NotImplementedError("getfactor of this complex factor (LDL' form, or an ordering that split a (Re, Im) pair)")
        Rp, Ri, Rx = F.fac.get_factor()
        n = F.n
        cpos = np.repeat(np.arange(2 * n, dtype=np.int64), np.diff(Rp))
        P2 = F.real_perm
        ev = (P2[cpos] % 2) == 0                            # the column that carries the real part of its pair
        rpos, cp_, xx = Ri[ev], cpos[ev], Rx[ev]
        re = (P2[rpos] % 2) == 0
        key = rpos // 2 + (cp_ // 2) * n                    # entries of one complex l_ij share (pair of the row, pair of the column)
        uniq = np.unique(key)
        val = np.zeros(uniq.size, dtype=np.complex128)
        np.add.at(val, np.searchsorted(uniq, key), np.where(re, xx, 1j * xx))
        dg = (uniq % n) == (uniq // n)
        val[dg] = val[dg].real                              # (the entry between the two halves of a pair is zero up to rounding)
        return spmatrix(val, uniq % n, uniq // n, (n, n), "z")
    Lp, Li, Lx = F.fac.get_factor()
    return spmatrix.from_ccs(F.fac.n, F.fac.n, Lp, Li, Lx)
