"""TEST INFRASTRUCTURE ONLY -- a test double for `kvxopt.cholmod`, used solely by
tests/golden/make_goldens.py inside the build container.

The reference's src/C/cholmod.c wraps SuiteSparse CHOLMOD, which is absent from this image, so the
module cannot be built.  To let the reference's OWN python (misc.kkt_chol2 sparse branch,
coneprog.conelp) run end to end for fixture generation, this file offers the same functions on
kvxopt's own matrix/spmatrix types, backed by the CPU oracle (oracle/kvx_oracle.c: plain-C up-looking
Cholesky pinned on the reference's documented known answers).  It is never imported by kvxopt_amd/ and
never shipped as part of the product.  Semantics follow cholmod.c:244-985 (file:line cited per function).
"""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from oracle.kvx_oracle import OracleChol  # noqa: E402

from kvxopt import matrix, spmatrix  # noqa: E402  (the reference's own types)

options = {}


class _F:
    def __init__(self, o):
        self.o = o
        self.numeric = False


def _ccs(A):
    cp, ri, v = A.CCS
    return (np.array(list(cp), dtype=np.int64), np.array(list(ri), dtype=np.int64), np.array(list(v), dtype=float))


def symbolic(A, p=None, uplo="L"):                 # cholmod.c:244-291
    cp, ri, v = _ccs(A)
    perm = None if p is None else np.array(list(p), dtype=np.int64)
    return _F(OracleChol(A.size[0], cp, ri, uplo, perm))


def numeric(A, F):                                 # cholmod.c:322-398 (documented behaviour: raise if not PD)
    F.o.factorize(_ccs(A)[2])
    F.numeric = True


def solve(F, B, sys=0, nrhs=-1, ldB=0, offsetB=0):  # cholmod.c:429-499
    if not F.numeric:
        raise ValueError("called with symbolic factor")
    n = F.o.n
    if nrhs < 0:
        nrhs = B.size[1]
    if n == 0 or nrhs == 0:
        return
    if ldB == 0:
        ldB = max(1, B.size[0])
    buf = np.array(list(B), dtype=float)
    for r in range(nrhs):
        col = buf[offsetB + r * ldB: offsetB + r * ldB + n].copy()
        F.o.solve(col, sys)
        buf[offsetB + r * ldB: offsetB + r * ldB + n] = col
    B[:] = matrix(buf, B.size)


def spsolve(F, B, sys=0):                          # cholmod.c:524-587
    if not F.numeric:
        raise ValueError("called with symbolic factor")
    n = F.o.n
    m, ncol = B.size
    if n == 0 or ncol == 0:
        return spmatrix([], [], [], (m, ncol))
    D = np.array(matrix(B)).reshape(m, ncol, order="F").copy(order="F")
    for j in range(ncol):
        col = D[:, j].copy()
        F.o.solve(col, sys)
        D[:, j] = col
    I, J = np.nonzero(D)
    return spmatrix(D[I, J].tolist(), I.tolist(), J.tolist(), (m, ncol))


def linsolve(A, B, p=None, uplo="L", nrhs=-1, ldB=0, offsetB=0):   # cholmod.c:618-753
    F = symbolic(A, p, uplo)
    numeric(A, F)
    solve(F, B, 0, nrhs, ldB, offsetB)


def splinsolve(A, B, p=None, uplo="L"):            # cholmod.c:774-881
    F = symbolic(A, p, uplo)
    numeric(A, F)
    return spsolve(F, B, 0)


def diag(F):                                       # cholmod.c:900-945
    return matrix(F.o.diag())
