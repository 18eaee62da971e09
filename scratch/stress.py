#!/usr/bin/env python3
"""Randomised parity stress (GPU vs the CPU oracle): patterns, sizes, nrhs, all solve kinds.  Not part of the test suite."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
from oracle.kvx_oracle import OracleChol
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
def run(n, cp, ri, vx, tag):
    global worst
    nrhs = int(rng.integers(1, 5))
    F = Factor(n, cp, ri); F.factorize(vx)
    O = OracleChol(n, cp, ri, "L", F.perm()); O.factorize(vx)
    B = rng.standard_normal((n, nrhs))
    for sys_ in (0, 4, 5):
        X = np.asfortranarray(B.copy()); Xo = np.asfortranarray(B.copy())
        F.solve(X, sys=sys_); O.solve(Xo, sys=sys_)
        e = np.abs(X - Xo).max() / max(np.abs(Xo).max(), 1e-300)
        worst = max(worst, e)
        assert e < 1e-9, (tag, n, sys_, e)
    e = np.abs(F.diag() - O.diag()).max() / np.abs(O.diag()).max()
    assert e < 1e-11, (tag, n, "diag", e)
    info = F.info()
    return info["nsuper"], info["nlevels"], info["max_front"]
t0 = time.time(); cases = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    kind = it % 5
    if kind == 0:
        n = int(rng.integers(40, 3000)); dens = float(rng.uniform(0.5, 8.0)) / n
        M = sp.random(n, n, dens, random_state=int(rng.integers(1 << 30)), format="csc")
        S = sp.tril((M @ M.T + sp.eye(n) * (1.0 + rng.uniform())).tocsc()).tocsc(); S.sort_indices()
        r = run(n, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data, "rand")
    elif kind == 1:
        r = run(*workloads.laplacian_2d(int(rng.integers(3, 140)), int(rng.integers(3, 140))), "lap2d")
    elif kind == 2:
        r = run(*workloads.laplacian_3d(int(rng.integers(3, 22))), "lap3d")
    elif kind == 3:
        r = run(*workloads.stencil21_2d(int(rng.integers(6, 70))), "st21")
    else:   # arrow / banded + dense rows: many children per front
        n = int(rng.integers(100, 1500)); bw = int(rng.integers(1, 6))
        A = sp.diags([np.full(n - k, -1.0 / (k + 1)) for k in range(1, bw + 1)], [-k for k in range(1, bw + 1)], shape=(n, n), format="lil")
        nd = int(rng.integers(1, 40))
        for i in range(n - nd, n):
            A[i, :i] = -rng.uniform(0.0, 1.0, i) * (rng.uniform(0, 1, i) < 0.3) / n
        A = sp.csc_matrix(A); A = A + sp.eye(n) * (bw + 2.0); A = sp.tril(A).tocsc(); A.sort_indices()
        r = run(n, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data, "arrow")
    cases += 1
print("stress ok: %d cases, worst rel diff %.2e, %.1f s" % (cases, worst, time.time() - t0))
