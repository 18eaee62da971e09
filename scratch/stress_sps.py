#!/usr/bin/env python3
"""Randomised stress of spsolve's reach-restricted forward systems against the dense column blocks (bit for bit) on random
patterns, grids and 3-D grids, LL' and LDL' factors, with and without the leaf-subtree walks.  Not part of the test suite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); cases = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    kind = it % 4
    if kind == 0:
        g, h = int(rng.integers(20, 260)), int(rng.integers(20, 260)); n, cp, ri, vx = workloads.laplacian_2d(g, h); tag = "2d %dx%d" % (g, h)
    elif kind == 1:
        g = int(rng.integers(8, 34)); n, cp, ri, vx = workloads.laplacian_3d(g); tag = "3d %d" % g
    elif kind == 2:
        n = int(rng.integers(200, 6000)); R = sp.random(n, n, density=min(0.5, 3.0 / n), random_state=rng, format="csc")
        M = sp.tril(R + R.T).tocsc(); M = (M + sp.diags(np.asarray(abs(M + M.T).sum(axis=1)).ravel() + 1.0)).tocsc(); M.sort_indices()
        cp, ri, vx = M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.copy(); tag = "rand %d" % n
    else:
        g = int(rng.integers(30, 120)); n, cp, ri, vx = workloads.stencil21_2d(g); tag = "st21 %d" % g
    F = Factor(n, cp, ri, opts={"supernodal": int(rng.choice([0, 2]))})
    F.factorize(vx)
    ncol = int(rng.integers(1, 200))
    Bp, Bi, Bx = [0], [], []
    for j in range(ncol):
        cnt = int(rng.integers(0, 5))
        rows = np.sort(rng.choice(n, size=min(cnt, n), replace=False))
        Bi.extend(int(r) for r in rows); Bx.extend(rng.standard_normal(rows.size)); Bp.append(len(Bi))
    Bp, Bi, Bx = np.array(Bp, dtype=np.int64), np.array(Bi, dtype=np.int64), np.array(Bx)
    for sys_ in (4, 2):
        os.environ.pop("KVX_SPSOLVE_DENSE", None)
        a = F.spsolve(ncol, Bp, Bi, Bx, sys=sys_)
        os.environ["KVX_SPSOLVE_DENSE"] = "1"
        b = F.spsolve(ncol, Bp, Bi, Bx, sys=sys_)
        for u, v in zip(a, b):
            assert np.array_equal(u, v), (tag, sys_)
    cases += 1
    print("%-14s n=%-7d ncol=%-4d nnz(X)=%d ok" % (tag, n, ncol, a[0][-1]), flush=True)
print("%d cases, %.1f s" % (cases, time.time() - t0))
