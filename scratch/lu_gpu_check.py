import sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo")
from kvxopt_amd import klu
from kvxopt_amd.base import spmatrix, matrix

def to_sp(S):
    return sp.csc_matrix((S.values, S.rowind, S.colptr), shape=S.size)

def check(name, A, nrhs=3):
    n = A.size[0]
    As = to_sp(A)
    t0 = time.time(); Fs = klu.symbolic(A); t1 = time.time(); Fn = klu.numeric(A, Fs); t2 = time.time()
    print(name, "n", n, "sym", Fs.sym.info(), "num", Fn.num.info(), "t_sym %.3f t_num %.3f" % (t1 - t0, t2 - t1), flush=True)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((n, nrhs))
    for tr in ("N", "T"):
        x = np.asfortranarray(b.copy())
        klu.solve(A, Fs, Fn, x, trans=tr)
        M = As if tr == "N" else As.T
        r = M @ x - b
        print("  trans", tr, "resid_inf %.3e" % np.abs(r).max(), "rel %.3e" % (np.linalg.norm(r) / np.linalg.norm(b)), flush=True)
    L, U, P, Q, R, F, r = klu.get_numeric(A, Fs, Fn)
    res = to_sp(R) @ to_sp(P) @ As @ to_sp(Q) - (to_sp(L) @ to_sp(U) + to_sp(F))
    print("  |RPAQ - LU - F|_1 = %.3e" % abs(res).sum(axis=0).max(), "lnz", L.values.size, "unz", U.values.size, "r", r, flush=True)
    Lc = to_sp(L); Uc = to_sp(U)
    assert abs(sp.triu(Lc, 1)).sum() == 0 and abs(sp.tril(Uc, -1)).sum() == 0
    assert np.allclose(Lc.diagonal(), 1.0)
    if n <= 2100:
        d = klu.get_det(A, Fs, Fn)
        sgn, logdet = np.linalg.slogdet(As.toarray())
        print("  det", d, "numpy", sgn, logdet, flush=True)
    # refactor with changed values, same pattern
    A2 = spmatrix.from_ccs(n, n, A.colptr, A.rowind, A.values * (1.0 + 0.1 * rng.random(A.values.size)))
    klu.numeric(A2, Fs, Fn)
    x = np.asfortranarray(b.copy()); klu.solve(A2, Fs, Fn, x)
    print("  refactor resid %.3e" % np.abs(to_sp(A2) @ x - b).max(), "passes", Fn.num.info()["passes"], flush=True)

V = [2, 3, 3, -1, 4, 4, -3, 1, 2, 2, 6, 1]; I = [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4]; J = [0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 4, 4]
A = spmatrix(V, I, J)
B = matrix(np.arange(5.0), (5, 1))
klu.linsolve(A, B)
print("doc linsolve:", np.array(B._a).ravel(), "expected [5.26e-02 -3.51e-02 3.00 5.48 -1.86]")
check("doc5", A, 2)
for nm in sys.argv[1:]:
    z = np.load("/root/repo/tests/golden/%s.npz" % nm)
    check(nm, spmatrix.from_ccs(int(z["n"]), int(z["n"]), z["colptr"], z["rowind"], z["values"]))
