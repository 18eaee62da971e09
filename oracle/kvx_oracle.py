"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front-end of the CPU oracle.

Restates what the reference computes on the KKT factor/solve hot path (orthant 'l' cone) and, for
the widening row SURVEY 8(f)4, the Nesterov-Todd scaling operations of the 'q' and 's' blocks;
every function cites the reference lines it follows.
Never imported by kvxopt_amd/ (the product path fails loudly without its HIP
library instead of falling back to this).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def _ptr_i(a):
    return a.ctypes.data_as(_i64p)


def _ptr_d(a):
    return a.ctypes.data_as(_f64p)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libkvxoracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libkvxoracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.kvxo_chol_analyze.restype = ctypes.c_void_p
        L.kvxo_chol_analyze.argtypes = [ctypes.c_int64, _i64p, _i64p, ctypes.c_int, _i64p]
        L.kvxo_chol_free.argtypes = [ctypes.c_void_p]
        L.kvxo_chol_factorize.argtypes = [ctypes.c_void_p, _f64p]
        L.kvxo_chol_factorize.restype = ctypes.c_int
        L.kvxo_chol_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, _f64p, ctypes.c_int64, ctypes.c_int64]
        L.kvxo_chol_solve.restype = ctypes.c_int
        for f in ("kvxo_chol_n", "kvxo_chol_lnz", "kvxo_chol_minor"):
            getattr(L, f).restype = ctypes.c_int64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.kvxo_chol_flops.restype = ctypes.c_double
        L.kvxo_chol_flops.argtypes = [ctypes.c_void_p]
        L.kvxo_chol_get_perm.argtypes = [ctypes.c_void_p, _i64p]
        L.kvxo_chol_get_parent.argtypes = [ctypes.c_void_p, _i64p]
        L.kvxo_chol_get_L.argtypes = [ctypes.c_void_p, _i64p, _i64p, _f64p]
        L.kvxo_chol_diag.argtypes = [ctypes.c_void_p, _f64p]
        L.kvxo_klu_factor.restype = ctypes.c_void_p
        L.kvxo_klu_factor.argtypes = [ctypes.c_int64, _i64p, _i64p, _f64p, _i64p, ctypes.c_double]
        L.kvxo_klu_free.argtypes = [ctypes.c_void_p]
        L.kvxo_klu_singular.argtypes = [ctypes.c_void_p]
        L.kvxo_klu_singular.restype = ctypes.c_int
        for f in ("kvxo_klu_lnz", "kvxo_klu_unz"):
            getattr(L, f).restype = ctypes.c_int64
            getattr(L, f).argtypes = [ctypes.c_void_p]
        L.kvxo_klu_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, _f64p, ctypes.c_int64, ctypes.c_int64]
        L.kvxo_klu_solve.restype = ctypes.c_int
        L.kvxo_klu_extract.argtypes = [ctypes.c_void_p, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p]
        L.kvxo_klu_det.argtypes = [ctypes.c_void_p]
        L.kvxo_klu_det.restype = ctypes.c_double
        L.kvxo_spmv.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, _f64p,
                                ctypes.c_double, _f64p, ctypes.c_double, _f64p]
        L.kvxo_rowscale.argtypes = [ctypes.c_int64, _i64p, _i64p, _f64p, _f64p, _f64p]
        L.kvxo_syrk_partial.argtypes = [ctypes.c_int64, ctypes.c_int64, _i64p, _i64p, _f64p,
                                        _i64p, _i64p, _f64p]
        _LIB = L
    return _LIB


class OracleChol:
    """P*A*P' = L*L' on the CPU (cholmod.c:244-499 semantics; see kvx_oracle.c)."""

    def __init__(self, n, colptr, rowind, uplo="L", perm=None):
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        self.rowind = np.ascontiguousarray(rowind, dtype=np.int64)
        self.n = int(n)
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
        self._h = lib().kvxo_chol_analyze(self.n, _ptr_i(self.colptr), _ptr_i(self.rowind),
                                          ord(uplo), None if p is None else _ptr_i(p))
        if not self._h:
            raise ValueError("p is not a valid permutation")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kvxo_chol_free(self._h)
            self._h = None

    @property
    def lnz(self):
        return lib().kvxo_chol_lnz(self._h)

    @property
    def flops(self):
        return lib().kvxo_chol_flops(self._h)

    @property
    def minor(self):
        return lib().kvxo_chol_minor(self._h)

    def perm(self):
        p = np.empty(self.n, dtype=np.int64)
        lib().kvxo_chol_get_perm(self._h, _ptr_i(p))
        return p

    def parent(self):
        p = np.empty(self.n, dtype=np.int64)
        lib().kvxo_chol_get_parent(self._h, _ptr_i(p))
        return p

    def factorize(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.size == self.colptr[-1]
        st = lib().kvxo_chol_factorize(self._h, _ptr_d(v))
        if st:
            raise ArithmeticError(int(self.minor))

    def solve(self, B, sys=0):
        """B: (n,) or (n, nrhs) Fortran-ordered float64, overwritten."""
        B2 = B.reshape(self.n, -1, order="F") if B.ndim == 1 else B
        assert B2.flags.f_contiguous or B2.shape[1] == 1
        st = lib().kvxo_chol_solve(self._h, int(sys), _ptr_d(B2), B2.shape[1], max(1, self.n))
        if st == 1:
            raise ArithmeticError("singular matrix")
        if st == 2:
            raise ValueError("called with symbolic factor")
        if st == 3:
            raise ValueError("invalid value for sys")
        return B

    def diag(self):
        d = np.empty(self.n)
        lib().kvxo_chol_diag(self._h, _ptr_d(d))
        return d

    def L(self):
        Lp = np.empty(self.n + 1, dtype=np.int64)
        Li = np.empty(self.lnz, dtype=np.int64)
        Lx = np.empty(self.lnz)
        lib().kvxo_chol_get_L(self._h, _ptr_i(Lp), _ptr_i(Li), _ptr_d(Lx))
        return Lp, Li, Lx


def spmv(trans, m, n, colptr, rowind, values, x, y, alpha=1.0, beta=0.0):
    """y := alpha*op(A)*x + beta*y, CCS A (sparse.c:1073-1104)."""
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowind, dtype=np.int64)
    v = np.ascontiguousarray(values, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().kvxo_spmv(ord(trans), m, n, _ptr_i(cp), _ptr_i(ri), _ptr_d(v), alpha, _ptr_d(x), beta, _ptr_d(y))
    return y


def atda(ml, n, Gp, Gi, Gx, di, Sp, Si):
    """S = G' diag(di)^2 G on S's fixed lower pattern: row scale then partial syrk
    (misc.py:1418-1419,1451 -> sparse.c:1260-1283, 2176-2198)."""
    Gp = np.ascontiguousarray(Gp, dtype=np.int64)
    Gi = np.ascontiguousarray(Gi, dtype=np.int64)
    Gx = np.ascontiguousarray(Gx, dtype=np.float64)
    Sp = np.ascontiguousarray(Sp, dtype=np.int64)
    Si = np.ascontiguousarray(Si, dtype=np.int64)
    di = np.ascontiguousarray(di, dtype=np.float64)
    Gs = np.empty_like(Gx)
    lib().kvxo_rowscale(n, _ptr_i(Gp), _ptr_i(Gi), _ptr_d(Gx), _ptr_d(di), _ptr_d(Gs))
    Sx = np.zeros(Si.size)
    lib().kvxo_syrk_partial(ml, n, _ptr_i(Gp), _ptr_i(Gi), _ptr_d(Gs), _ptr_i(Sp), _ptr_i(Si), _ptr_d(Sx))
    return Gs, Sx


# --- Nesterov-Todd scaling, orthant ('l') blocks only ------------------------------
def compute_scaling_l(s, z):
    """misc.py:284-287: d = sqrt(s./z), di = d**-1, lmbda = sqrt(s.*z)."""
    d = np.sqrt(s / z)
    return d, d ** -1, np.sqrt(s * z)


def update_scaling_l(d, lmbda, s, z):
    """misc.py:444-464, in place like the reference: s,z := sqrt; d := d.*s./z;
    di := d**-1; lmbda := s.*z.  Returns di."""
    np.sqrt(s, out=s)
    np.sqrt(z, out=z)
    d *= s
    d /= z
    lmbda[:] = s * z
    return d ** -1


def scale_l(x, d, di, inverse="N"):
    """misc_solvers.c:132-141 / misc.py:74-82: x := d.*x ('N') or di.*x ('I'),
    every column; trans is irrelevant for a diagonal W."""
    w = d if inverse == "N" else di
    x *= w.reshape(-1, *([1] * (x.ndim - 1)))
    return x


def scale2_l(lmbda, x, inverse="N"):
    """misc_solvers.c:287-298: x := x./lmbda ('N') or x.*lmbda ('I')."""
    if inverse == "N":
        x /= lmbda.reshape(-1, *([1] * (x.ndim - 1)))
    else:
        x *= lmbda.reshape(-1, *([1] * (x.ndim - 1)))
    return x


def sprod_l(x, y):
    """misc_solvers.c:662-669: x := x.*y."""
    x *= y
    return x


def sinv_l(x, y):
    """misc_solvers.c:793-800: x := x./y."""
    x /= y
    return x


def ssqr_l(y):
    """misc.py:951-952: x := y.*y."""
    return y * y


def sdot_l(x, y):
    """misc_solvers.c:1018: sum x_i*y_i."""
    return float(np.dot(x, y))


def max_step_l(x):
    """misc_solvers.c:1065-1071: max_i(-x_i)  (-inf... the reference starts from
    -FLT_MAX; for ml>0 this is max(-x))."""
    return float(np.max(-x)) if x.size else -np.finfo(np.float64).max


# --- second-order-cone ('q') blocks, one cone at a time (numpy vectors) ----------------
def _jnrm2(x):
    """misc.py:848-856: sqrt(x0 - |x1|) * sqrt(x0 + |x1|)."""
    a = np.linalg.norm(x[1:])
    return np.sqrt(x[0] - a) * np.sqrt(x[0] + a)


def compute_scaling_q(s, z):
    """misc.py:290-352 for one cone: returns v, beta, lmbda with (beta (2 v v' - J)) z = lmbda."""
    aa, bb = _jnrm2(s), _jnrm2(z)
    beta = np.sqrt(aa / bb)
    cc = np.sqrt((np.dot(s, z) / aa / bb + 1.0) / 2.0)
    v = -z / bb
    v[0] *= -1.0
    v += s / aa
    v *= 1.0 / 2.0 / cc
    v[0] += 1.0
    v *= 1.0 / np.sqrt(2.0 * v[0])
    lm = np.empty_like(s)
    lm[0] = cc
    dd = 2 * cc + s[0] / aa + z[0] / bb
    lm[1:] = (cc + z[0] / bb) / dd / aa * s[1:] + (cc + s[0] / aa) / dd / bb * z[1:]
    lm *= np.sqrt(aa * bb)
    return v, float(beta), lm


def scale_q(x, v, beta, inverse="N"):
    """misc_solvers.c:144-186 for one cone and one column: x := beta (2 v v' - J) x, or its inverse (1/beta)(2 J v v' J - J) x."""
    x = x.copy()
    if inverse == "I":
        x[0] *= -1.0
    w = np.dot(v, x)
    x[0] *= -1.0
    x += 2.0 * w * v
    if inverse == "I":
        x[0] *= -1.0
        return x / beta
    return x * beta


def sprod_q(x, y):
    """misc_solvers.c:671-700 for one cone: (y o x)_0 = y'x, (y o x)_1 = y0 x1 + x0 y1."""
    out = y[0] * x + x[0] * y
    out[0] = np.dot(x, y)
    return out


def max_step_q(x):
    """misc_solvers.c:1073-1085 for one cone: |x1| - x0."""
    return float(np.linalg.norm(x[1:]) - x[0])


# --- semidefinite ('s') blocks, one block at a time (m x m numpy matrices) --------------
def compute_scaling_s(S, Z):
    """misc.py:354-419 for one block: r, rti, lmbda with r' Z r = r^-1 S r^-T = diag(lmbda); lmbda = singular values of
    Lz' Ls in descending order.  (The signs of the columns of r are those of numpy's SVD -- any choice is a valid scaling.)"""
    Ls, Lz = np.linalg.cholesky(S), np.linalg.cholesky(Z)
    U, lm, _ = np.linalg.svd(Lz.T @ Ls)
    r = np.linalg.solve(Lz.T, U) * np.sqrt(lm)
    rti = (Lz @ U) / np.sqrt(lm)
    return r, rti, lm


def update_scaling_s(r, rti, Ls, Lz):
    """misc.py:582-634 for one block: returns the new r, rti, lmbda (Ls, Lz: the blocks of s and z on entry)."""
    U, lm, Vt = np.linalg.svd(Lz.T @ Ls)
    return (r @ Ls @ Vt.T) / np.sqrt(lm), (rti @ Lz @ U) / np.sqrt(lm), lm


def scale_s(X, r, rti, trans="N", inverse="N"):
    """misc_solvers.c:188-240 for one block and one column: the lower triangle of r' Xs r ('N','N'), r Xs r' ('T','N'),
    rti Xs rti' ('N','I'), rti' Xs rti ('T','I') with Xs the symmetric matrix stored in the lower triangle of X; the strict
    upper triangle of X is returned unchanged."""
    Xs = np.tril(X) + np.tril(X, -1).T
    R = r if inverse == "N" else rti
    Y = (R @ Xs @ R.T) if (inverse == "N") == (trans == "T") else (R.T @ Xs @ R)
    return np.tril(Y) + np.triu(X, 1)


def scale2_s(lm, X, inverse="N"):
    """misc_solvers.c:343-397 for one block: X(i, j) / (sqrt(l_i) sqrt(l_j)) ('N') or times it ('I'), every entry."""
    c = np.outer(np.sqrt(lm), np.sqrt(lm))
    return X / c if inverse == "N" else X * c


def sprod_s(X, Y, diag="N"):
    """misc_solvers.c:700-770 for one block: the lower triangle of (Y X + X Y) / 2 (X, Y symmetric from their lower triangles;
    diag 'D': Y = diag(y)); the strict upper triangle of X unchanged."""
    Xs = np.tril(X) + np.tril(X, -1).T
    Ys = np.diag(Y) if diag == "D" else np.tril(Y) + np.tril(Y, -1).T
    return np.tril(0.5 * (Ys @ Xs + Xs @ Ys)) + np.triu(X, 1)


def sinv_s(X, y):
    """misc_solvers.c:845-882 for one block: lower triangle of X divided entrywise by (y_i + y_j) / 2."""
    c = 0.5 * (y[:, None] + y[None, :])
    return np.tril(X / c) + np.triu(X, 1)


def sdot_s(X, Y):
    """misc_solvers.c:1029-1046 for one block: trace inner product of the symmetric matrices in the lower triangles."""
    return float(np.sum(np.diag(X) * np.diag(Y)) + 2.0 * np.sum(np.tril(X, -1) * np.tril(Y, -1)))


def max_step_s(X):
    """misc_solvers.c:1086-1160 for one block: -lambda_min of the symmetric matrix in the lower triangle, and its eigenvalues
    (ascending)."""
    ev = np.linalg.eigvalsh(np.tril(X) + np.tril(X, -1).T)
    return float(-ev[0]), ev


def pack_s(X):
    """misc_solvers.c:412-468 for one block: the lower triangle by columns, off-diagonal entries times sqrt(2) (the diagonal is
    divided and multiplied by sqrt(2) as the reference does)."""
    m = X.shape[0]
    out = []
    for k in range(m):
        col = X[k:, k].copy()
        col[0] /= np.sqrt(2.0)
        out.append(col)
    return np.concatenate(out) * np.sqrt(2.0)


class OracleKLU:
    """Left-looking LU with threshold partial pivoting and row scaling (oracle/klu_oracle.c): the checker for the
    kvxopt.klu path (reference: src/C/klu.c:141-198, 444-449, 760-822).  Raises ArithmeticError like klu.c:172-174."""

    def __init__(self, n, colptr, rowind, values, Q=None, tol=1e-3):
        self.n = int(n)
        cp = np.ascontiguousarray(colptr, dtype=np.int64)
        ri = np.ascontiguousarray(rowind, dtype=np.int64)
        v = np.ascontiguousarray(values, dtype=np.float64)
        q = None if Q is None else np.ascontiguousarray(Q, dtype=np.int64)
        self._h = lib().kvxo_klu_factor(self.n, _ptr_i(cp), _ptr_i(ri), _ptr_d(v), None if q is None else _ptr_i(q), tol)
        if not self._h:
            raise MemoryError()
        if lib().kvxo_klu_singular(self._h):
            raise ArithmeticError("singular matrix")

    def solve(self, B, trans="N"):
        X = np.array(B, dtype=np.float64, order="F", copy=True)
        X2 = X.reshape(self.n, -1, order="F")
        rc = lib().kvxo_klu_solve(self._h, 0 if trans == "N" else 1, _ptr_d(X2), X2.shape[1], self.n)
        if rc:
            raise ArithmeticError("singular matrix")
        return X

    def extract(self):
        n = self.n
        lnz = lib().kvxo_klu_lnz(self._h) + n
        unz = lib().kvxo_klu_unz(self._h) + n
        Lp = np.empty(n + 1, np.int64); Li = np.empty(lnz, np.int64); Lx = np.empty(lnz)
        Up = np.empty(n + 1, np.int64); Ui = np.empty(unz, np.int64); Ux = np.empty(unz)
        P = np.empty(n, np.int64); Q = np.empty(n, np.int64); Rs = np.empty(n)
        lib().kvxo_klu_extract(self._h, _ptr_i(Lp), _ptr_i(Li), _ptr_d(Lx), _ptr_i(Up), _ptr_i(Ui), _ptr_d(Ux),
                               _ptr_i(P), _ptr_i(Q), _ptr_d(Rs))
        return (Lp, Li, Lx), (Up, Ui, Ux), P, Q, Rs

    def det(self):
        return lib().kvxo_klu_det(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kvxo_klu_free(self._h)
            self._h = None


# ---- host supernodal multifrontal Cholesky on all cores (oracle/kvx_supernodal.c): the CPU baseline beside the GPU numbers
_SLIB = None


def supernodal_lib():
    """libkvxsupernodal.so (OpenMP + the OpenBLAS inside scipy); built on demand.  Raises OSError when it cannot be built."""
    global _SLIB
    if _SLIB is None:
        so = os.path.join(_HERE, "libkvxsupernodal.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libkvxsupernodal.so"])
        L = ctypes.CDLL(so)
        L.kvxs_analyze.restype = ctypes.c_void_p
        L.kvxs_analyze.argtypes = [ctypes.c_int64, _i64p, _i64p, _i64p, ctypes.c_int64, _i64p, _i64p, _i64p, _i64p]
        L.kvxs_free.argtypes = [ctypes.c_void_p]
        L.kvxs_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.kvxs_factorize.argtypes = [ctypes.c_void_p, _f64p]
        L.kvxs_factorize.restype = ctypes.c_int
        L.kvxs_solve.argtypes = [ctypes.c_void_p, _f64p, ctypes.c_int64, ctypes.c_int64]
        L.kvxs_solve.restype = ctypes.c_int
        L.kvxs_minor.argtypes = [ctypes.c_void_p]
        L.kvxs_minor.restype = ctypes.c_int64
        _SLIB = L
    return _SLIB


class OracleSupernodal:
    """Supernodal multifrontal P A P' = L L' on the host cores, on the supernodes and permutation of a kvxopt_amd.chol.Factor
    (`structure` = (perm, super, front rowptr, front rowidx, parent), all host-side analysis results): the role CHOLMOD's
    supernodal factorisation plays for the reference (cholmod.c:362-364, 483).  A must be the lower-triangular CCS."""

    def __init__(self, n, colptr, rowind, structure, threads=None):
        perm, sup, rp, ri, parent = [np.ascontiguousarray(a, dtype=np.int64) for a in structure]
        self.n = int(n)
        self.cp = np.ascontiguousarray(colptr, dtype=np.int64)
        self.ri = np.ascontiguousarray(rowind, dtype=np.int64)
        col = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.cp))
        if np.any(self.ri < col):
            raise ValueError("OracleSupernodal expects the lower triangle")
        L = supernodal_lib()
        self._h = L.kvxs_analyze(self.n, _ptr_i(self.cp), _ptr_i(self.ri), _ptr_i(perm), len(sup) - 1, _ptr_i(sup), _ptr_i(rp),
                                 _ptr_i(ri), _ptr_i(parent))
        if not self._h:
            raise MemoryError
        self.threads = max(1, min(int(threads or len(os.sched_getaffinity(0)) or 1), 32))
        L.kvxs_set_threads(self._h, self.threads)

    @classmethod
    def from_factor(cls, n, colptr, rowind, F, threads=None):
        sup, nrows, parent, level = F.supernodes()
        rp, ri = F.front_rows()
        return cls(n, colptr, rowind, (F.perm(), sup, rp, ri, parent), threads)

    def __del__(self):
        if getattr(self, "_h", None):
            supernodal_lib().kvxs_free(self._h)
            self._h = None

    def factorize(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        rc = supernodal_lib().kvxs_factorize(self._h, _ptr_d(v))
        if rc == 1:
            raise ArithmeticError(int(supernodal_lib().kvxs_minor(self._h)))
        if rc:
            raise MemoryError

    def solve(self, B):
        """B: column-major (n x nrhs) float64, overwritten with the solution of A X = B."""
        B2 = B.reshape(self.n, -1, order="F") if B.ndim == 1 else B
        assert B2.flags.f_contiguous and B2.dtype == np.float64
        rc = supernodal_lib().kvxs_solve(self._h, _ptr_d(B2), B2.shape[1], max(self.n, 1))
        if rc:
            raise ArithmeticError("singular matrix")
