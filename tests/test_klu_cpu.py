"""CPU-side checks of the sparse LU path (kvxopt.klu API): the oracle (oracle/klu_oracle.c) is pinned on the
reference's documented known answers (doc/source/spsolvers.rst:333-345, 420-439) and on the identities the
reference's tests assert (tests/test_sparse_solvers.py:214-323), then the host analysis (matching, front plan)
and the argument validation of the klu mirror are checked -- none of it needs a GPU; the numeric phase must fail
loudly here."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from kvxopt_amd import _lib, klu
from kvxopt_amd.base import matrix, spmatrix
from kvxopt_amd.lu import LuSymbolic
from oracle.kvx_oracle import OracleKLU

# spsolvers.rst:322-345
DOC_V = [2, 3, 3, -1, 4, 4, -3, 1, 2, 2, 6, 1]
DOC_VB = [4, 3, 3, -1, 4, 4, -3, 1, 2, 2, 6, 2]
DOC_I = [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4]
DOC_J = [0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 4, 4]


def doc_csc(V):
    A = sp.csc_matrix((np.array(V, float), (DOC_I, DOC_J)), shape=(5, 5))
    A.sort_indices()
    return A


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n = int(z["n"])
    return n, z["colptr"], z["rowind"], z["values"], sp.csc_matrix((z["values"], z["rowind"], z["colptr"]), shape=(n, n))


def test_oracle_doc_linsolve_known_answer():
    A = doc_csc(DOC_V)
    F = OracleKLU(5, A.indptr, A.indices, A.data)
    x = F.solve(np.arange(5.0))
    assert np.allclose(x, [5.26e-02, -3.51e-02, 3.00e+00, 5.48e+00, -1.86e+00], rtol=5e-3)      # spsolvers.rst:340-345
    assert np.allclose(A @ x, np.arange(5.0), atol=1e-13)


def test_oracle_doc_three_solves_known_answer():
    """x = A^-T B^-1 A^-1 1 (spsolvers.rst:420-439)."""
    A, B = doc_csc(DOC_V), doc_csc(DOC_VB)
    FA = OracleKLU(5, A.indptr, A.indices, A.data)
    FB = OracleKLU(5, B.indptr, B.indices, B.data)
    x = FA.solve(FB.solve(FA.solve(np.ones(5))), "T")
    assert np.allclose(x, [5.81e-01, -2.37e-01, 1.63e+00, 8.07e+00, -1.31e-01], rtol=5e-3)


def test_oracle_determinant_and_identity_doc():
    A = doc_csc(DOC_V)
    F = OracleKLU(5, A.indptr, A.indices, A.data)
    assert abs(F.det() - np.linalg.det(A.toarray())) < 1e-10 * 114         # test_sparse_solvers.py:303-310
    (L, U, P, Q, Rs) = F.extract()
    Ls = sp.csc_matrix((L[2], L[1], L[0]), shape=(5, 5)); Us = sp.csc_matrix((U[2], U[1], U[0]), shape=(5, 5))
    lhs = sp.diags(1 / Rs) @ A.tocsr()[P, :].tocsc()[:, Q]
    assert abs(lhs - Ls @ Us).max() < 1e-15                                 # R P A Q = L U + F, F = 0


@pytest.mark.parametrize("name", ["bp_800", "ACTIVSg2000"])
def test_oracle_on_reference_matrices(golden_dir, name):
    """test_sparse_solvers.py:239-259 style residual checks (places=7) + an independent SuperLU cross-check."""
    n, cp, ri, v, A = load(golden_dir, name)
    q = None
    if name == "ACTIVSg2000":                                               # a fill-reducing column order keeps the oracle fast
        q = spla.splu(A, permc_spec="COLAMD").perm_c.astype(np.int64)
    F = OracleKLU(n, cp, ri, v, Q=q)
    b = np.random.default_rng(3).standard_normal((n, 3))
    for tr in "NT":
        x = F.solve(b, tr)
        M = A if tr == "N" else A.T
        assert np.abs(M @ x - b).max() < 1e-7
        xs = spla.splu(M.tocsc()).solve(b)
        assert np.abs(x - xs).max() <= 1e-8 * max(1.0, np.abs(xs).max())


def test_oracle_singular():
    A = sp.csc_matrix(np.array([[1.0, 2.0], [2.0, 4.0]]))
    with pytest.raises(ArithmeticError):
        OracleKLU(2, A.indptr, A.indices, A.data)


# ---- host analysis (no GPU) -----------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bp_800", "ACTIVSg2000", "bcsstk13"])
def test_matching_puts_nonzeros_on_the_diagonal(golden_dir, name):
    n, cp, ri, v, A = load(golden_dir, name)
    S = LuSymbolic(n, cp, ri, v)
    rowfor = S.matching()
    assert sorted(rowfor) == list(range(n))
    d = np.abs(np.asarray(A[rowfor, np.arange(n)]).ravel())
    assert d.min() > 0.0
    inf = S.info()
    assert inf["structurally_singular"] == 0 and inf["n"] == n and inf["nnz"] == v.size
    # the matching maximises the product of the row-scaled diagonal: it cannot be worse than the identity's when that is zero-free
    rs = np.abs(A).max(axis=1).toarray().ravel()
    if np.all(A.diagonal() != 0):
        assert np.log(d / rs[rowfor]).sum() >= np.log(np.abs(A.diagonal()) / rs).sum() - 1e-9


def test_matching_is_optimal_on_small_random_matrices():
    from itertools import permutations
    rng = np.random.default_rng(7)
    for _ in range(30):
        n = 5
        D = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.6)
        D[np.arange(n), rng.permutation(n)] += 1.0                          # structurally nonsingular
        A = sp.csc_matrix(D); A.sort_indices()
        rowfor = LuSymbolic(n, A.indptr, A.indices, A.data).matching()
        W = np.abs(D) / np.abs(D).max(axis=1, keepdims=True)
        best = max(np.prod([W[p[j], j] for j in range(n)]) for p in permutations(range(n)))
        got = np.prod([W[rowfor[j], j] for j in range(n)])
        assert got >= best * (1 - 1e-12)


@pytest.mark.parametrize("name,nblocks", [("bp_800", 492), ("ACTIVSg2000", 1)])      # (ACTIVSg2000 stores explicit zeros: with them it is irreducible)
def test_block_triangular_form(golden_dir, name, nblocks):
    """KLU's BTF: strongly connected components of the matched matrix, numbered so that R P A Q is block UPPER triangular;
    the count is checked against SciPy's strongly connected components of the same matched matrix."""
    from scipy.sparse.csgraph import connected_components
    n, cp, ri, v, A = load(golden_dir, name)
    S = LuSymbolic(n, cp, ri, v)
    nb, nlev, blk = S.btf()
    rowfor = S.matching()
    M = A.tocsr()[rowfor, :].tocoo()                       # M(j, j) = A(rowfor[j], j)
    assert nb == nblocks == connected_components(M.tocsr(), directed=True, connection="strong")[0]
    assert np.all(blk[M.row] <= blk[M.col])                 # nothing below the block diagonal
    assert 1 <= nlev <= 64 and sorted(set(blk)) == list(range(nb))


def test_deep_block_chains_fall_back_to_one_block(golden_dir):
    """A triangular matrix is n singleton blocks in one dependency chain: factoring it as one block keeps the solves parallel."""
    n, cp, ri, v, A = load(golden_dir, "bcsstk13")          # stored lower triangle, read as a general matrix
    nb, nlev, blk = LuSymbolic(n, cp, ri, v).btf()
    assert nb == 1 and nlev == 1 and not blk.any()


def test_structurally_singular_pattern_is_flagged():
    A = sp.csc_matrix(np.array([[1.0, 1.0, 0], [1.0, 1.0, 0], [1.0, 1.0, 0]]))       # column 2 empty
    S = LuSymbolic(3, A.indptr, A.indices, A.data)
    assert S.info()["structurally_singular"] == 1
    assert sorted(S.matching()) == [0, 1, 2]


def test_klu_argument_validation():
    A = spmatrix(DOC_V, DOC_I, DOC_J)
    with pytest.raises(TypeError):
        klu.symbolic(spmatrix([1.0, 2.0], [0, 1], [0, 1], (2, 3)))          # klu.c:251-252
    with pytest.raises(TypeError):
        klu.symbolic(matrix(np.eye(2)))
    with pytest.raises(ValueError):
        klu.symbolic(spmatrix([], [], [], (0, 0)))                          # klu.c:256-260
    Fs = klu.symbolic(A)
    assert Fs.name == "KLU SYM D FACTOR"
    with pytest.raises(TypeError):
        klu.numeric(A, "not a factor")                                       # klu.c:322
    with pytest.raises(ValueError):
        klu.numeric(spmatrix([1.0, 2.0], [0, 1], [0, 1], (5, 5)), Fs)        # other pattern
    with pytest.raises(TypeError):
        klu.solve(A, Fs, "x", matrix(np.ones(5)))
    with pytest.raises(ValueError):
        LuSymbolic(2, np.array([0, 2, 1]), np.array([0, 1]), None)          # colptr not monotone
    with pytest.raises(ValueError):
        LuSymbolic(2, np.array([0, 1, 2]), np.array([0, 5]), None)          # row index out of range


def test_numeric_fails_loudly_without_gpu():
    if _lib.lib().kvx_device_count() > 0:
        pytest.skip("a GPU is visible")
    A = spmatrix(DOC_V, DOC_I, DOC_J)
    Fs = klu.symbolic(A)
    with pytest.raises(RuntimeError):
        klu.numeric(A, Fs)
    with pytest.raises(RuntimeError):
        klu.linsolve(A, matrix(np.arange(5.0)))
