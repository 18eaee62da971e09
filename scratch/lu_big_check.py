import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, "/root/repo")
from kvxopt_amd import klu, _lib
from kvxopt_amd.base import spmatrix

def convdiff(g, seed=0):
    """Unsymmetric 5-point convection-diffusion operator on a g x g grid with random local wind."""
    rng = np.random.default_rng(seed)
    n = g * g
    idx = np.arange(n).reshape(g, g)
    I, J, V = [idx.ravel()], [idx.ravel()], [np.full(n, 4.0) + 0.1 * rng.random(n)]
    for (a, b, s) in ((idx[1:, :], idx[:-1, :], 1), (idx[:-1, :], idx[1:, :], -1), (idx[:, 1:], idx[:, :-1], 1), (idx[:, :-1], idx[:, 1:], -1)):
        w = rng.standard_normal(a.size) * 0.8
        I.append(a.ravel()); J.append(b.ravel()); V.append(-1.0 + s * w)
    A = sp.csc_matrix((np.concatenate(V), (np.concatenate(I), np.concatenate(J))), shape=(n, n)); A.sort_indices()
    return A

for g in [int(a) for a in sys.argv[1:]]:
    As = convdiff(g); n = As.shape[0]
    A = spmatrix.from_ccs(n, n, As.indptr, As.indices, As.data)
    t0 = time.perf_counter(); Fs = klu.symbolic(A); t1 = time.perf_counter(); Fn = klu.numeric(A, Fs); t2 = time.perf_counter()
    print("grid", g, "n", n, "sym %.3fs first numeric %.3fs" % (t1 - t0, t2 - t1), Fs.sym.info(), Fn.num.info(), flush=True)
    vals_d = _lib.DeviceBuffer.from_array(A.values)
    t0 = time.perf_counter()
    for _ in range(3): Fn.num.refactor_dev(vals_d.ptr, A.values.size)
    t_ref = (time.perf_counter() - t0) / 3
    b = np.random.default_rng(1).standard_normal((n, 2))
    for tr in "NT":
        x = np.asfortranarray(b.copy())
        t0 = time.perf_counter(); klu.solve(A, Fs, Fn, x, trans=tr); ts = time.perf_counter() - t0
        M = As if tr == "N" else As.T
        print("  trans", tr, "resid rel %.3e" % (np.linalg.norm(M @ x - b) / np.linalg.norm(b)), "solve %.1f ms" % (ts * 1e3), flush=True)
    t0 = time.perf_counter(); lu = spla.splu(As); xs = lu.solve(b); tc = time.perf_counter() - t0
    x = np.asfortranarray(b.copy()); klu.solve(A, Fs, Fn, x)
    print("  refactor %.1f ms; scipy SuperLU factor+solve %.1f ms; |x - x_superlu| %.2e" % (t_ref * 1e3, tc * 1e3, np.abs(x - xs).max()), flush=True)
