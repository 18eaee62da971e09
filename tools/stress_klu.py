"""Random unsymmetric sparse systems through kvxopt_amd.klu against SciPy's SuperLU (run on the GPU box).

    python3 tools/stress_klu.py [count] [seed]

Every case: a random pattern (n = 20 .. 3000, 2 .. 12 entries per row, a fraction of the diagonal removed so that rows must be
interchanged), factor, solve A x = b and A' x = b, refactor with perturbed values on the recorded pivot sequence, solve again;
the solutions must agree with SuperLU's to a tolerance scaled by its own residual, singular matrices must raise on both sides.
"""
import os, sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvxopt_amd import klu
from kvxopt_amd.base import spmatrix


def case(rng):
    n = int(rng.integers(20, 3000))
    per_row = int(rng.integers(2, 13))
    dens = min(1.0, per_row / n)
    M = sp.random(n, n, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csc")
    d = rng.standard_normal(n) * (1.0 + 3.0 * rng.random())
    keep = rng.random(n) > rng.choice([0.0, 0.05, 0.3])
    M = (M + sp.diags(np.where(keep, d, 0.0))).tocsc()
    if rng.random() < 0.5:                                     # a band on top: larger fronts
        M = (M + sp.diags(rng.standard_normal(n - 1), 1) + sp.diags(rng.standard_normal(n - 1), -1)).tocsc()
    M.sort_indices()
    return M


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = 0.0
    nsing = 0
    for it in range(count):
        M = case(rng)
        n = M.shape[0]
        A = spmatrix.from_ccs(n, n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.copy())
        b = rng.standard_normal((n, 2))
        try:
            lu = spla.splu(M)
            ref_ok = np.all(np.isfinite(lu.solve(b)))
        except RuntimeError:
            ref_ok = False
        try:
            Fs = klu.symbolic(A)
            Fn = klu.numeric(A, Fs)
        except ArithmeticError:
            nsing += 1
            if ref_ok and np.linalg.cond(M.toarray()) < 1e10:
                raise SystemExit("case %d (n = %d): singular here, regular for SuperLU" % (it, n))
            continue
        if not ref_ok:
            continue
        for vals in (M.data, M.data * (1.0 + 0.05 * rng.standard_normal(M.nnz))):
            Mv = sp.csc_matrix((vals, M.indices, M.indptr), shape=M.shape)
            Av = spmatrix.from_ccs(n, n, M.indptr.astype(np.int64), M.indices.astype(np.int64), vals.copy())
            if vals is not M.data:
                try:
                    klu.numeric(Av, Fs, Fn)
                    luv = spla.splu(Mv)
                except (ArithmeticError, RuntimeError):
                    break
            else:
                luv = lu
            for tran in "NT":
                x = np.asfortranarray(b.copy())
                klu.solve(Av, Fs, Fn, x, trans=tran)
                xs = luv.solve(b, trans=tran)
                Mt = Mv if tran == "N" else Mv.T
                rs = np.linalg.norm(Mt @ xs - b) / np.linalg.norm(b)
                r = np.linalg.norm(Mt @ x - b) / np.linalg.norm(b)
                worst = max(worst, r / max(rs, 1e-16))
                if not (r <= max(1e-9, 1e4 * rs)):
                    raise SystemExit("case %d (n = %d, nnz = %d, trans %s): residual %.2e, SuperLU %.2e" % (it, n, M.nnz, tran, r, rs))
    print("%d cases, %d singular on both sides, worst residual ratio to SuperLU %.1f" % (count, nsing, worst))


if __name__ == "__main__":
    main()
