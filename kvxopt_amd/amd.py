"""Drop-in for `kvxopt.amd` (src/C/amd.c:131-223): `order(A, uplo='L') -> p`, a fill-reducing permutation of a
symmetric sparse matrix, as an 'i' matrix -- what `cholmod.symbolic(A, p)` and `misc` callers expect.

Not SuiteSparse AMD: the permutation comes from this package's own host analysis (nested dissection on level
structures + halo-aware minimum degree on the leaves, elimination-tree postorder; csrc/ordering.cpp, symbolic.cpp), the
one the GPU Cholesky uses when no `p` is given.  Any valid permutation satisfies the reference's contract; its quality
(fill) is what the tests compare.  Host-only: no GPU needed.  `options` is accepted and validated like amd.c:57-110
(the AMD control parameters have no meaning here and are ignored).
"""
import numpy as np

from . import base
from .base import matrix, spmatrix
from .chol import Factor

options = {}


def order(A, uplo="L"):
    for k, v in options.items():
        if isinstance(k, str) and not isinstance(v, (int, float)):
            raise ValueError("invalid value for AMD parameter: %-.20s" % k)
    if not (isinstance(A, spmatrix) or hasattr(A, "CCS")):
        raise TypeError("A must be a square sparse matrix")
    m, n, cp, ri, v = base._as_ccs(A)
    if m != n:
        raise TypeError("A must be a square sparse matrix")
    if uplo not in ("L", "U"):
        raise ValueError("possible values of uplo are: 'L', 'U'")
    F = Factor(n, cp, ri, uplo)             # host analysis only (reads the `uplo` triangle, amd.c:161-203)
    return matrix(F.perm(), (n, 1), tc="i")
