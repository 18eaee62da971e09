"""Drop-in for `kvxopt.amd` (src/C/amd.c:131-223): `order(A, uplo='L') -> p`, a fill-reducing permutation of a
symmetric sparse matrix, as an 'i' matrix -- what `cholmod.symbolic(A, p)` and `misc` callers expect.

Not SuiteSparse AMD but the same algorithm family, written for this package (csrc/amd_order.cpp): approximate minimum
degree on the quotient graph -- elements instead of explicit fill, approximate external degrees, element absorption, mass
elimination of indistinguishable variables -- followed by the elimination-tree postorder of the analysis.  Fill is within a
few per cent of SciPy/SuperLU's MMD on the reference's own test matrices (tests/test_symbolic.py).  Host-only: no GPU
needed.  `options` is accepted and validated like amd.c:57-110 (the AMD control parameters are not used).
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
    F = Factor(n, cp, ri, uplo, None, {"ordering": 3})   # host analysis only (reads the `uplo` triangle, amd.c:161-203)
    return matrix(F.perm(), (n, 1), tc="i")
