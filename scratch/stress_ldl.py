#!/usr/bin/env python3
"""Random check of the LDL' factor semantics (options['supernodal'] = 0) against dense numpy: all sys codes, user permutations,
getfactor.  Not part of the test suite."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp
from kvxopt_amd import cholmod
from kvxopt_amd.base import matrix, spmatrix
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
cholmod.options["supernodal"] = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = int(rng.integers(1, 400))
    M = sp.random(n, n, min(1.0, float(rng.uniform(0.5, 6.0)) / n), random_state=int(rng.integers(1 << 30)), format="csc")
    S = (M @ M.T + sp.eye(n) * (0.5 + rng.uniform())).toarray()
    I, J = np.nonzero(np.tril(S))
    A = spmatrix(S[I, J], I, J, (n, n))
    p = rng.permutation(n) if it % 2 else None
    F = cholmod.symbolic(A, p=None if p is None else matrix(p, tc="i"))
    cholmod.numeric(A, F)
    perm = F.fac.perm()
    if p is not None and F.fac.info()["nsuper"] > 0:
        pass
    nrhs = int(rng.integers(1, 5))
    B = rng.standard_normal((n, nrhs))
    got = {}
    for s in range(9):
        X = matrix(B.copy(order="F"))
        cholmod.solve(F, X, sys=s)
        got[s] = X.a.reshape(n, nrhs).copy()
    Mx = cholmod.getfactor(F).todense()
    D = np.diag(np.diag(Mx)); L = np.tril(Mx, -1) + np.eye(n)
    P = np.eye(n)[perm]
    e0 = np.abs(L @ D @ L.T - S[np.ix_(perm, perm)]).max() / np.abs(S).max()
    assert e0 < 1e-12, ("factor", n, e0)
    ops = {0: S, 1: L @ D @ L.T, 2: L @ D, 3: D @ L.T, 4: L, 5: L.T, 6: D, 7: P.T, 8: P}
    for s, Mop in ops.items():
        ref = np.linalg.solve(Mop, B)
        e = np.abs(got[s] - ref).max() / max(np.abs(ref).max(), 1e-300)
        worst = max(worst, e)
        assert e < 1e-8, (n, s, e)
cholmod.options.clear()
print("ldl stress ok, worst rel diff %.2e" % worst)
