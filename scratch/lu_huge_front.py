import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo")
from kvxopt_amd import klu
from kvxopt_amd.base import spmatrix
rng = np.random.default_rng(0)
n = 5200
D = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.3)
D[np.arange(n), np.arange(n)] = rng.standard_normal(n) * (rng.random(n) < 0.5)       # half the diagonal zero: pivoting needed
A = sp.csc_matrix(D); A.sort_indices()
M = spmatrix.from_ccs(n, n, A.indptr, A.indices, A.data)
t0 = time.time(); Fs = klu.symbolic(M); Fn = klu.numeric(M, Fs); print("factor %.2fs" % (time.time() - t0), Fn.num.info(), flush=True)
b = rng.standard_normal((n, 2))
for tr in "NT":
    x = np.asfortranarray(b.copy()); klu.solve(M, Fs, Fn, x, trans=tr)
    r = (D if tr == "N" else D.T) @ x - b
    print(tr, "resid", np.abs(r).max(), "x", np.abs(x).max(), flush=True)
xs = np.linalg.solve(D, b)
x = np.asfortranarray(b.copy()); klu.solve(M, Fs, Fn, x)
print("vs LAPACK", np.abs(x - xs).max() / np.abs(xs).max())
