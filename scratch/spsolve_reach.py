#!/usr/bin/env python3
"""spsolve(sys = 4) with sparse right-hand sides: reach-restricted sweep against the dense column blocks (KVX_SPSOLVE_DENSE=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor

for g, h, ncol in ((250, 200, 2048), (1000, 1000, 512)):
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    F = Factor(n, cp, ri)
    F.factorize(v + 0.0)
    rng = np.random.default_rng(1)
    Bp = np.arange(0, 2 * ncol + 1, 2, dtype=np.int64)
    Bi = np.sort(rng.integers(0, n, size=(ncol, 2)), axis=1).reshape(-1).astype(np.int64)
    Bx = rng.standard_normal(2 * ncol)
    for mode in ("reach", "dense"):
        if mode == "dense":
            os.environ["KVX_SPSOLVE_DENSE"] = "1"
        else:
            os.environ.pop("KVX_SPSOLVE_DENSE", None)
        F.spsolve(64, Bp[:65], Bi[:128], Bx[:128], sys=4)
        t = time.perf_counter()
        Xp, Xi, Xx = F.spsolve(ncol, Bp, Bi, Bx, sys=4)
        dt = time.perf_counter() - t
        print("n=%d ncol=%d %s: %.1f ms, nnz(X)=%d (%.2f %% of n*ncol)" % (n, ncol, mode, 1e3 * dt, Xp[-1], 100.0 * Xp[-1] / (n * ncol)), flush=True)
