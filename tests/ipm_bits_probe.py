"""Helper of test_kkt_gpu.py::test_fused_iteration_is_bitwise: solves a few problems with the device-resident interior-point
drivers and prints a digest of every returned array (run once with KVX_LP_UNFUSED=1, once without)."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import lp as kvx_lp, workloads          # noqa: E402
from kvxopt_amd.base import spmatrix                    # noqa: E402


def digest(sol):
    out = {"status": sol["status"], "iterations": sol["iterations"]}
    for k in ("x", "s", "z", "y"):
        v = sol.get(k)
        if v is not None:
            out[k] = hashlib.sha256(np.ascontiguousarray(v, dtype=np.float64).tobytes()).hexdigest()
    for k in ("gap", "primal objective", "dual objective", "primal infeasibility", "dual infeasibility"):
        out[k] = None if sol.get(k) is None else float(sol[k]).hex()
    return out


res = {}
for name, (gx, gy) in (("grid30x20", (30, 20)), ("grid61x47", (61, 47))):
    P = workloads.lp_grid(gx, gy)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    res[name] = digest(kvx_lp.conelp(P["c"], G, P["h"]))
# long rows and columns (the 16-lane forms of the fused kernels): random rows with up to 12 entries over a box
rng = np.random.default_rng(11)
n, mr = 60, 240
rows, cols, vals = [], [], []
for i in range(mr):
    js = rng.choice(n, size=int(rng.integers(1, 13)), replace=False)
    rows += [i] * len(js); cols += list(js); vals += list(rng.standard_normal(len(js)))
for j in range(n):                                      # -1 <= x_j <= 1
    rows += [mr + 2 * j, mr + 2 * j + 1]; cols += [j, j]; vals += [1.0, -1.0]
ml = mr + 2 * n
import scipy.sparse as sp                               # noqa: E402
Gs = sp.csc_matrix((vals, (rows, cols)), shape=(ml, n)); Gs.sort_indices()
x0 = rng.uniform(-0.5, 0.5, n)
h = Gs @ x0 + rng.uniform(0.1, 1.0, ml)
h[mr:] = 1.0
G = spmatrix.from_ccs(ml, n, Gs.indptr.astype(np.int64), Gs.indices.astype(np.int64), Gs.data)
res["random_long_rows"] = digest(kvx_lp.conelp(rng.standard_normal(n), G, h))
assert np.diff(Gs.indptr).max() > 8 and np.bincount(Gs.indices).max() > 8

# a QP on the same constraints (coneqp uses KKTChol2Dev.solve / factor with H = P)
P = workloads.lp_grid(25, 18)
n = P["n"]
G = spmatrix.from_ccs(P["ml"], n, P["Gp"], P["Gi"], P["Gx"])
H = spmatrix.from_ccs(n, n, np.arange(n + 1), np.arange(n), np.full(n, 0.5))
res["qp25x18"] = digest(kvx_lp.coneqp(H, P["c"], G, P["h"]))
print(json.dumps(res))
