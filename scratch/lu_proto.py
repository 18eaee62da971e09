"""Prototype (CPU, numpy): static-structure LU with in-supernode threshold pivoting -- how robust is it on the
reference's KLU test matrices?  Reads the .mtx data files of the reference checkout (prototype only)."""
import sys, numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import min_weight_full_bipartite_matching
sys.path.insert(0, "/root/repo")
from kvxopt_amd.chol import Factor

def read_mtx(fn):
    I, J, V = [], [], []; size = None
    for row in open(fn):
        if row.startswith('%'): continue
        if size is None: size = list(map(int, row.split())); continue
        a = row.split(); I.append(int(a[0]) - 1); J.append(int(a[1]) - 1); V.append(float(a[2]))
    return sp.csc_matrix((V, (I, J)), shape=(size[0], size[1]))

def run(name, tol=1e-3, weighted=True):
    A = read_mtx("/root/reference/tests/" + name); n = A.shape[0]
    A.sum_duplicates(); A.eliminate_zeros()
    rs = np.abs(A).max(axis=1).toarray().ravel()
    As = (sp.diags(1 / rs) @ A).tocsc()
    C = As.tocoo()
    wts = 1.0 - np.log(np.abs(C.data)) if weighted else np.ones(C.nnz)
    w = sp.csr_matrix((wts, (C.row, C.col)), shape=(n, n))
    r, c = min_weight_full_bipartite_matching(w)
    rowfor = np.empty(n, int); rowfor[c] = r
    M = As.tocsr()[rowfor, :].tocsc()           # M[j,j] = As[rowfor[j], j]
    pat = (abs(M) + abs(M.T) + sp.identity(n)).tocsc(); pat.sort_indices()
    Lp = sp.tril(pat).tocsc(); Lp.sort_indices()
    F = Factor(n, Lp.indptr, Lp.indices)
    perm = F.perm(); sup, nrows, parent, level = F.supernodes()
    info = F.info()
    D = M.toarray()[np.ix_(perm, perm)]
    D0 = D.copy()
    nswap = 0; minpiv = np.inf; growth = 0.0
    rowp = np.arange(n)
    for s in range(len(sup) - 1):
        j0, j1 = sup[s], sup[s + 1]
        for j in range(j0, j1):
            col = np.abs(D[j:j1, j])
            imax = int(np.argmax(col))
            if col[0] >= tol * col[imax]: imax = 0
            if imax:
                D[[j, j + imax], :] = D[[j + imax, j], :]; rowp[[j, j + imax]] = rowp[[j + imax, j]]; nswap += 1
            p = D[j, j]
            minpiv = min(minpiv, abs(p))
            if p == 0.0:
                print(name, "ZERO pivot at", j, "supernode size", j1 - j0); return
            rows = j + 1 + np.nonzero(D[j + 1:, j])[0]
            cols = j + 1 + np.nonzero(D[j, j + 1:])[0]
            D[rows, j] /= p
            if rows.size and cols.size:
                D[np.ix_(rows, cols)] -= np.outer(D[rows, j], D[j, cols])
        growth = max(growth, np.abs(D[j0:j1, :]).max())
    L = np.tril(D, -1) + np.eye(n); U = np.triu(D)
    # NOTE: swaps were applied LAPACK-style to full rows here (single P)
    res = np.abs(D0[rowp, :] - L @ U).sum(axis=0).max()
    b = np.random.default_rng(0).standard_normal(n)
    y = np.linalg.solve(L, b[rowp]) if n <= 5000 else None
    x = np.linalg.solve(U, y)
    r1 = np.abs(D0 @ x - b).max()
    print(f"{name}: n={n} nsuper={len(sup)-1} lnz={info['lnz']} swaps={nswap} minpiv={minpiv:.2e} maxU={growth:.2e} "
          f"|PM-LU|_1={res:.2e} resid_inf={r1:.2e} |x|={np.abs(x).max():.2e}")

if __name__ == "__main__":
    for nm in sys.argv[1:]:
        run(nm)
