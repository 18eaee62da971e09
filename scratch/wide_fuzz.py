"""randomised check of the rhs-major solves against the CPU oracle: patterns, sizes, nrhs, systems, LL' / LDL' views"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
os.environ["KVX_WIDE_FROM"] = "8"
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from oracle.kvx_oracle import OracleChol
_lib.require_device()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    kind = rng.integers(0, 4)
    if kind == 0:
        g, h = int(rng.integers(3, 90)), int(rng.integers(3, 90))
        n, cp, ri, v = workloads.laplacian_2d(g, h); name = "grid %dx%d" % (g, h)
    elif kind == 1:
        g = int(rng.integers(3, 20))
        n, cp, ri, v = workloads.laplacian_3d(g); name = "cube %d" % g
    elif kind == 2:
        g = int(rng.integers(5, 50))
        n, cp, ri, v = workloads.stencil21_2d(g); name = "stencil21 %d" % g
    else:
        n = int(rng.integers(1, 2500)); dens = float(rng.choice([0.0005, 0.002, 0.01, 0.05]))
        M = sp.random(n, n, dens, random_state=int(rng.integers(1 << 30)), format="csc")
        L = sp.tril((M @ M.T + sp.eye(n) * 2.0).tocsc()).tocsc(); L.sort_indices()
        cp, ri, v = L.indptr.astype(np.int64), L.indices.astype(np.int64), L.data; name = "random %d %.4f" % (n, dens)
    sn = int(rng.choice([0, 2]))
    F = Factor(n, cp, ri, opts={"supernodal": sn})
    F.factorize(v)
    O = OracleChol(n, cp, ri, "L", F.perm())
    O.factorize(v)
    nrhs = int(rng.integers(8, 200))
    B = rng.standard_normal((n, nrhs))
    for sysc in ((0, 4, 5) if sn == 2 else (0, 1, 2, 3, 4, 5)):
        X = np.asfortranarray(B.copy()); F.solve(X, sys=sysc)
        if sn == 2:
            Xo = np.asfortranarray(B.copy()); O.solve(Xo, sys=sysc)
        else:                                    # LDL' view: compare with the library's own single-rhs solves
            Xo = np.empty_like(X)
            for j in range(0, nrhs, max(1, nrhs // 5)):
                xj = B[:, j].copy(); F.solve(xj, sys=sysc); Xo[:, j] = xj
            idx = list(range(0, nrhs, max(1, nrhs // 5)))
            X, Xo = X[:, idx], Xo[:, idx]
        err = np.abs(X - Xo).max() / max(np.abs(Xo).max(), 1e-300)
        if not err < 1e-10:
            bad += 1
            print("MISMATCH", name, "n", n, "nrhs", nrhs, "sys", sysc, "supernodal", sn, "err %.2e" % err, flush=True)
print("fuzz done, mismatches:", bad)
