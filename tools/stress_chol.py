#!/usr/bin/env python3
"""Randomised residual check of the Cholesky path (factorize, solve, the one-enqueue form, several right-hand sides) on systems the
fixed test-suite does not hold: grids of random shape, random sparse SPD matrices, arrow / banded patterns.  Prints one line per
failure and a summary; exit code 1 on any failure.  Usage: python tools/stress_chol.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import _lib, workloads          # noqa: E402
from kvxopt_amd.chol import Factor              # noqa: E402


def lower(A):
    L = sp.tril(sp.csc_matrix(A)).tocsc(); L.sort_indices()
    return A.shape[0], L.indptr.astype(np.int64), L.indices.astype(np.int64), L.data.astype(np.float64)


def case(rng, kind):
    if kind == 0:
        g, h = int(rng.integers(3, 260)), int(rng.integers(3, 260))
        return "lap2d %dx%d" % (g, h), workloads.laplacian_2d(g, h)
    if kind == 1:
        g = int(rng.integers(3, 34))
        return "lap3d %d" % g, workloads.laplacian_3d(g)
    if kind == 2:
        g = int(rng.integers(6, 110))
        return "stencil21 %d" % g, workloads.stencil21_2d(g)
    if kind == 3:                                   # random sparse SPD: B B' + shift
        n = int(rng.integers(20, 6000)); per = int(rng.integers(1, 6))
        B = sp.random(n, n, density=min(1.0, per / n), random_state=int(rng.integers(1 << 30)), format="csc")
        A = (B @ B.T + sp.identity(n) * (1.0 + rng.random())).tocsc()
        return "random n=%d per=%d" % (n, per), lower(A)
    if kind == 4:                                   # arrow + band: a few dense rows at the end
        n = int(rng.integers(50, 4000)); bw = int(rng.integers(1, 40)); nd = int(rng.integers(1, 70))
        rows, cols, vals = [], [], []
        for d in range(1, bw + 1):
            i = np.arange(d, n); rows += list(i); cols += list(i - d); vals += list(rng.uniform(-1, 1, n - d) / bw)
        for r in range(n - nd, n):
            j = np.arange(0, r); rows += list(np.full(r, r)); cols += list(j); vals += list(rng.uniform(-1, 1, r) / n)
        S = sp.csc_matrix((vals, (rows, cols)), shape=(n, n)); S = S + S.T
        A = (S + sp.identity(n) * (3.0 + abs(S).sum(axis=1).max())).tocsc()
        return "arrow n=%d bw=%d dense=%d" % (n, bw, nd), lower(A)
    n = int(rng.integers(1, 40))                    # tiny dense
    M = rng.standard_normal((n, n)); A = sp.csc_matrix(M @ M.T + n * np.eye(n))
    return "dense n=%d" % n, lower(A)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    t0 = time.time()
    for it in range(cases):
        name, (n, cp, ri, v) = case(rng, it % 6)
        nrhs = int(rng.choice([1, 1, 2, 3, 5, 17]))
        B = np.asfortranarray(rng.standard_normal((n, nrhs)))
        F = Factor(n, cp, ri)
        F.factorize(v)
        X = B.copy(order="F"); F.solve(X)
        R = np.stack([workloads.sym_matvec(n, cp, ri, v, X[:, j]) for j in range(nrhs)], 1) - B
        anorm = np.abs(v).max() * 8
        res = np.abs(R).max() / (anorm * max(np.abs(X).max(), 1e-300) + np.abs(B).max())
        ok = res < 1e-12
        # the one-enqueue form, repeated (eager, capture, replay): bit for bit the two calls
        vd = _lib.DeviceBuffer.from_array(v); bd = _lib.DeviceBuffer(8 * n * nrhs)
        for rep in range(3):
            bd.upload(B.reshape(-1, order="F"))
            F.factorize_solve_dev(vd.ptr, bd.ptr, nrhs, n)
            X2 = bd.download(np.float64, n * nrhs).reshape(n, nrhs, order="F")
            ok = ok and np.array_equal(X2, X)
        if not ok:
            bad += 1
            print("FAIL %s nrhs=%d residual %.2e fused-equal %s" % (name, nrhs, res, np.array_equal(X2, X)), flush=True)
        if it % 20 == 19:
            print("%d cases, %d failures, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d failures" % (cases, bad))
    sys.exit(1 if bad else 0)


main()
