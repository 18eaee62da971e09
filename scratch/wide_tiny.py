import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from kvxopt_amd.chol import Factor
from kvxopt_amd import _lib
_lib.require_device()
rng = np.random.default_rng(0)
ok = True
for n in (1, 2, 5, 17, 33, 65, 130, 300):
    for dens in (1.0, 0.2):
        M = sp.random(n, n, dens, random_state=n, format="csc") if dens < 1 else sp.csc_matrix(rng.standard_normal((n, n)))
        S = (M @ M.T + sp.eye(n) * (n + 1.0)).tocsc()
        L = sp.tril(S).tocsc(); L.sort_indices()
        F = Factor(n, L.indptr.astype(np.int64), L.indices.astype(np.int64))
        F.factorize(L.data)
        for nr in (64, 100):
            B = rng.standard_normal((n, nr))
            X = np.asfortranarray(B.copy())
            F.solve(X)
            R = S @ X - B
            err = np.abs(R).max() / np.abs(B).max()
            if not err < 1e-10:
                ok = False
                print("n", n, "dens", dens, "nrhs", nr, "residual", err)
print("tiny cases", "OK" if ok else "FAILED")
