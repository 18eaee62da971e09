"""Pins the CPU oracle (oracle/kvx_oracle.c) before anything trusts it:
known answers from the reference's documentation (doc/source/spsolvers.rst:555-563, 580-585,
700-708, 759-772), dense numpy cross-checks, and the reference's own test matrices."""
import os

import numpy as np
import pytest

from oracle.kvx_oracle import OracleChol, atda, spmv

# spsolvers.rst:556: A = spmatrix([10,3,5,-2,5,2], [0,2,1,3,2,3], [0,0,1,1,2,3])
DOC_P = np.array([0, 2, 4, 5, 6])
DOC_I = np.array([0, 2, 1, 3, 2, 3])
DOC_V = np.array([10.0, 3, 5, -2, 5, 2])
DOC_DENSE = np.array([[10, 0, 3, 0], [0, 5, 0, -2], [3, 0, 5, 0], [0, -2, 0, 2.0]])


def test_doc_linsolve_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    X = np.asfortranarray(np.arange(8, dtype=float).reshape(4, 2, order="F"))
    F.solve(X)
    # printed to 3 significant digits at spsolvers.rst:560-563 / 705-708
    doc = np.array([[-1.46e-01, 4.88e-02], [1.33e+00, 4.00e+00], [4.88e-01, 1.17e+00], [2.83e+00, 7.50e+00]])
    assert np.allclose(X, doc, rtol=5e-3)
    assert np.allclose(DOC_DENSE @ X, np.arange(8, dtype=float).reshape(4, 2, order="F"), atol=1e-13)


def test_doc_inverse_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    X = np.asfortranarray(np.eye(4))
    F.solve(X)
    doc = np.array([[1.22e-01, 0, -7.32e-02, 0], [0, 3.33e-01, 0, 3.33e-01],
                    [-7.32e-02, 0, 2.44e-01, 0], [0, 3.33e-01, 0, 8.33e-01]])   # spsolvers.rst:582-585
    assert np.allclose(X, doc, atol=6e-4)


def test_doc_logdet_known_answer():
    F = OracleChol(4, DOC_P, DOC_I)
    F.factorize(DOC_V)
    assert abs(2.0 * np.sum(np.log(F.diag())) - 5.50533153593) < 1e-10     # spsolvers.rst:765


def _rand_spd(n, dens, seed):
    import scipy.sparse as sp
    M = sp.random(n, n, dens, random_state=seed, format="csc")
    S = (M @ M.T + sp.eye(n) * (1.0 + seed)).tocsc()
    L = sp.tril(S).tocsc()
    L.sort_indices()
    return S, L


@pytest.mark.parametrize("uplo", ["L", "U"])
def test_against_dense_cholesky(uplo):
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    S, L = _rand_spd(150, 0.04, 2)
    p = rng.permutation(150)
    T = L if uplo == "L" else sp.triu(S).tocsc()
    T.sort_indices()
    F = OracleChol(150, T.indptr, T.indices, uplo, p)
    F.factorize(T.data)
    Ld = np.linalg.cholesky(S.toarray()[np.ix_(p, p)])
    Lp, Li, Lx = F.L()
    Lo = np.zeros((150, 150))
    for j in range(150):
        Lo[Li[Lp[j]:Lp[j + 1]], j] = Lx[Lp[j]:Lp[j + 1]]
    assert np.abs(Lo - Ld).max() < 1e-12 * np.abs(Ld).max()
    assert F.lnz == np.count_nonzero(np.abs(Ld) > 1e-300)
    b = rng.standard_normal(150)
    for sys_, ref in ((4, np.linalg.solve(Ld, b)), (5, np.linalg.solve(Ld.T, b)), (7, b[p]), (1, np.linalg.solve(Ld @ Ld.T, b))):
        x = b.copy()
        F.solve(x, sys=sys_)
        assert np.allclose(x, ref, rtol=1e-10, atol=1e-12)
    x = b.copy(); F.solve(x, sys=7); F.solve(x, sys=8)
    assert np.array_equal(x, b)


def test_other_triangle_is_ignored_and_errors():
    # cholmod.c:137-157: entries in the other triangle are silently ignored
    import scipy.sparse as sp
    S, L = _rand_spd(40, 0.1, 5)
    full = S.tocsc(); full.sort_indices()
    junk = full.copy(); junk.data = junk.data.copy()
    up = junk.indices < np.repeat(np.arange(40), np.diff(junk.indptr))
    junk.data[up] = 1e9
    F = OracleChol(40, junk.indptr, junk.indices, "L")
    F.factorize(junk.data)
    b = np.ones(40); x = b.copy(); F.solve(x)
    assert np.linalg.norm(S @ x - b) < 1e-10
    with pytest.raises(ValueError):
        OracleChol(3, [0, 1, 2, 3], [0, 1, 2], "L", [0, 0, 1])
    # not positive definite -> ArithmeticError(minor)
    F = OracleChol(2, [0, 2, 3], [0, 1, 1], "L")
    with pytest.raises(ArithmeticError) as e:
        F.factorize([1.0, 2.0, 1.0])
    assert e.value.args[0] == 1
    with pytest.raises(ArithmeticError):
        F.solve(np.ones(2))
    with pytest.raises(ValueError):
        OracleChol(2, [0, 2, 3], [0, 1, 1]).solve(np.ones(2))     # symbolic factor


@pytest.mark.parametrize("name", ["bcsstk13", "bcsstk24"])
def test_reference_test_matrices(golden_dir, name):
    """Residual test in the style of tests/test_sparse_solvers.py:239-259 on the reference's own
    matrices (lower triangle as stored, uplo='L'); bcsstk13 has cond ~1.1e10."""
    from kvxopt_amd.workloads import sym_matvec
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n, cp, ri, v = int(z["n"]), z["colptr"], z["rowind"], z["values"]
    B = np.random.default_rng(13).standard_normal((n, 3))
    X = np.asfortranarray(B.copy())
    F = OracleChol(n, cp, ri, "L")
    F.factorize(v)
    F.solve(X)
    R = sym_matvec(n, cp, ri, v, X) - B
    nrmA = np.sqrt(2 * np.sum(v ** 2) - np.sum(v[ri == np.repeat(np.arange(n), np.diff(cp))] ** 2))
    backward = np.linalg.norm(R) / (nrmA * np.linalg.norm(X) + np.linalg.norm(B))
    assert backward < 1e-14
    assert np.linalg.norm(R) / np.linalg.norm(B) < 5e-8


def test_atda_and_spmv_against_dense():
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    G = sp.random(40, 12, 0.2, random_state=3, format="csc"); G.sort_indices()
    di = rng.uniform(0.5, 2, 40)
    Sd = (G.T @ sp.diags(di ** 2) @ G).toarray()
    pat = sp.tril((abs(G).T @ abs(G))).tocsc(); pat.sort_indices()
    Gs, Sx = atda(40, 12, G.indptr, G.indices, G.data, di, pat.indptr, pat.indices)
    ref = Sd[pat.indices, np.repeat(np.arange(12), np.diff(pat.indptr))]
    assert np.allclose(Sx, ref, rtol=1e-13, atol=1e-14)
    x = rng.standard_normal(12); y = rng.standard_normal(40)
    y2 = y.copy(); spmv("N", 40, 12, G.indptr, G.indices, G.data, x, y2, 2.0, -0.5)
    assert np.allclose(y2, 2 * (G @ x) - 0.5 * y)
    x2 = x.copy(); spmv("T", 40, 12, G.indptr, G.indices, G.data, y, x2, -1.0, 1.0)
    assert np.allclose(x2, x - G.T @ y)
