"""Prototype 2: static LU + forced supernode merges on pivot failure (dense emulation)."""
import sys, numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import min_weight_full_bipartite_matching, connected_components
sys.path.insert(0, "/root/repo")
from kvxopt_amd.chol import Factor
from lu_proto import read_mtx

def run(name, tol=1e-3, stol=1e-6, btf=False):
    A = read_mtx("/root/reference/tests/" + name); n = A.shape[0]
    A.sum_duplicates(); A.eliminate_zeros()
    rs = np.abs(A).max(axis=1).toarray().ravel()
    As = (sp.diags(1 / rs) @ A).tocsc()
    C = As.tocoo()
    w = sp.csr_matrix((1.0 - np.log(np.abs(C.data)), (C.row, C.col)), shape=(n, n))
    r, c = min_weight_full_bipartite_matching(w)
    rowfor = np.empty(n, int); rowfor[c] = r
    M = As.tocsr()[rowfor, :].tocsc()
    nb, lab = connected_components(M, directed=True, connection='strong')
    sizes = np.bincount(lab)
    print(f"{name}: BTF blocks={nb} largest={sizes.max()} singletons={(sizes==1).sum()}")
    Mf = M
    if btf:
        # keep only entries inside diagonal blocks for the factorisation
        Cc = M.tocoo(); keep = lab[Cc.row] == lab[Cc.col]
        Mf = sp.csc_matrix((Cc.data[keep], (Cc.row[keep], Cc.col[keep])), shape=(n, n))
    pat = (abs(Mf) + abs(Mf.T) + sp.identity(n)).tocsc(); pat.sort_indices()
    Lp = sp.tril(pat).tocsc(); Lp.sort_indices()
    F = Factor(n, Lp.indptr, Lp.indices)
    perm = F.perm(); sup, nrows, parent, level = F.supernodes()
    ns = len(sup) - 1
    Dm = Mf.toarray()[np.ix_(perm, perm)]
    uf = list(range(ns))
    def find(a):
        while uf[a] != a: uf[a] = uf[uf[a]]; a = uf[a]
        return a
    nmerge = 0
    while True:
        # groups: root of each union = topmost supernode (we always union child into parent: uf[child]=parent root)
        grp = {}
        for s in range(ns): grp.setdefault(find(s), []).append(s)
        # contracted tree
        gpar = {g: (find(parent[g]) if parent[g] >= 0 else -1) for g in grp}
        kids = {g: [] for g in grp}
        roots = []
        for g, p in gpar.items():
            (kids[p] if p >= 0 else roots).append(g)
        order = []   # list of (cols list) per group in postorder
        sys.setrecursionlimit(100000)
        def visit(g):
            for ch in sorted(kids[g]): visit(ch)
            cols = np.concatenate([np.arange(sup[s], sup[s + 1]) for s in sorted(grp[g])])   # members ascending = topological
            order.append((g, cols))
        for g in sorted(roots): visit(g)
        newp = np.concatenate([c for _, c in order])
        D = Dm[np.ix_(newp, newp)].copy(); D0 = D.copy()
        rowp = np.arange(n)
        pos = 0; fail = None; nswap = 0; maxL = 0.0
        for g, cols in order:
            j0, j1 = pos, pos + len(cols); pos = j1
            for j in range(j0, j1):
                col = np.abs(D[j:j1, j]); imax = int(np.argmax(col))
                if col[0] >= tol * col[imax]: imax = 0
                cmax = np.abs(D[j:, j]).max()
                if col[imax] == 0.0 or col[imax] < stol * cmax:
                    fail = g; break
                if imax:
                    D[[j, j + imax], :] = D[[j + imax, j], :]; rowp[[j, j + imax]] = rowp[[j + imax, j]]; nswap += 1
                p = D[j, j]
                rows = j + 1 + np.nonzero(D[j + 1:, j])[0]; cols_ = j + 1 + np.nonzero(D[j, j + 1:])[0]
                D[rows, j] /= p
                if rows.size: maxL = max(maxL, np.abs(D[rows, j]).max())
                if rows.size and cols_.size: D[np.ix_(rows, cols_)] -= np.outer(D[rows, j], D[j, cols_])
            if fail is not None: break
        if fail is None: break
        if gpar[fail] < 0:
            print("  singular: failure in a root group"); return
        uf[fail] = gpar[fail]; nmerge += 1
    L = np.tril(D, -1) + np.eye(n); U = np.triu(D)
    res = np.abs(D0[rowp, :] - L @ U).sum(axis=0).max()
    b = np.random.default_rng(0).standard_normal(n)
    x = np.linalg.solve(U, np.linalg.solve(L, b[rowp]))
    gs = max(len(c) for _, c in order)
    nnzLU = np.count_nonzero(D)
    print(f"  merges={nmerge} largest pivot block={gs} swaps={nswap} max|L|={maxL:.1e} nnz(L+U)={nnzLU} |PM-LU|_1={res:.2e} "
          f"resid={np.abs(D0 @ x - b).max():.2e} |x|={np.abs(x).max():.1e}")

if __name__ == "__main__":
    nm = sys.argv[1]
    for stol in (1e-3, 1e-6, 1e-10):
        for btf in (False, True):
            print("stol", stol, "btf", btf); run(nm, stol=stol, btf=btf)
