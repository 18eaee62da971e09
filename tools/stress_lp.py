#!/usr/bin/env python3
"""Randomised end-to-end check of the device-resident conelp: feasible, bounded LPs of random size and sparsity (rows with 1 .. 20
entries over a box) against scipy.optimize.linprog (HiGHS).  Usage: python tools/stress_lp.py [cases] [seed]"""
import os
import sys

import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import lp as kvx_lp             # noqa: E402
from kvxopt_amd.base import spmatrix            # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
bad = 0
for it in range(cases):
    n = int(rng.integers(5, 500)); mr = int(rng.integers(1, 5 * n)); mx = int(rng.choice([2, 4, 8, 20]))
    rows, cols, vals = [], [], []
    for i in range(mr):
        js = rng.choice(n, size=min(n, int(rng.integers(1, mx + 1))), replace=False)
        rows += [i] * len(js); cols += list(js); vals += list(rng.standard_normal(len(js)))
    for j in range(n):
        rows += [mr + 2 * j, mr + 2 * j + 1]; cols += [j, j]; vals += [1.0, -1.0]
    ml = mr + 2 * n
    G = sp.csc_matrix((vals, (rows, cols)), shape=(ml, n)); G.sort_indices()
    x0 = rng.uniform(-0.5, 0.5, n)
    h = G @ x0 + rng.uniform(0.05, 1.0, ml); h[mr:] = 1.0
    c = rng.standard_normal(n)
    sol = kvx_lp.conelp(c, spmatrix.from_ccs(ml, n, G.indptr.astype(np.int64), G.indices.astype(np.int64), G.data), h)
    ref = linprog(c, A_ub=G, b_ub=h, bounds=(None, None), method="highs")
    ok = sol["status"] == "optimal" and ref.status == 0 and abs(sol["primal objective"] - ref.fun) <= 1e-6 * max(1.0, abs(ref.fun)) \
        and (G @ sol["x"] - h).max() <= 1e-6
    if not ok:
        bad += 1
        print("FAIL n=%d ml=%d maxrow=%d: %s %r vs %r" % (n, ml, mx, sol["status"], sol["primal objective"], ref.fun), flush=True)
print("done: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
