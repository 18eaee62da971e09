"""GPU parity tests of the sparse LU path (kvxopt.klu API through the C ABI kvx_lu_*): the reference's own tests
(tests/test_sparse_solvers.py:214-323) restated for the four reference matrices, the documented known answers
(doc/source/spsolvers.rst:333-345, 420-439), parity with the CPU oracle (oracle/klu_oracle.c), singular inputs,
refactorisation, ldB/offsetB handling, and random matrices with zero diagonals that force pivoting and front merges.

Tolerance: floating point -- the HIP factorisation eliminates in another order than the oracle, so solutions are
compared to 1e-9 * max(1, |x|) and residuals to the reference's own bar (assertAlmostEqual, 7 places)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from kvxopt_amd import _lib, klu
from kvxopt_amd.base import matrix, spmatrix
from oracle.kvx_oracle import OracleKLU

pytestmark = pytest.mark.gpu

CASES = ["ACTIVSg2000", "bcsstk13", "bcsstk24", "bp_800"]          # test_sparse_solvers.py:29-30
DOC_V = [2, 3, 3, -1, 4, 4, -3, 1, 2, 2, 6, 1]
DOC_VB = [4, 3, 3, -1, 4, 4, -3, 1, 2, 2, 6, 2]
DOC_I = [0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4]
DOC_J = [0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 4, 4]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    _lib.require_device()


def to_sp(S):
    return sp.csc_matrix((S.values, S.rowind, S.colptr), shape=S.size)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n = int(z["n"])
    return spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])


def from_dense(D):
    A = sp.csc_matrix(D); A.sort_indices()
    return spmatrix.from_ccs(D.shape[0], D.shape[1], A.indptr, A.indices, A.data)


@pytest.mark.parametrize("name", CASES)
def test_lu_identity(golden_dir, name):
    """test_sparse_solvers.py:216-236: norm(R*P*A*Q - (L*U + F), 1) == 0 to 7 places."""
    A = load(golden_dir, name)
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    L, U, P, Q, R, F, r = klu.get_numeric(A, Fs, Fn)
    rho = abs(to_sp(R) @ to_sp(P) @ to_sp(A) @ to_sp(Q) - (to_sp(L) @ to_sp(U) + to_sp(F))).sum(axis=0).max()
    assert rho < 5e-8
    Ls, Us = to_sp(L), to_sp(U)
    assert abs(sp.triu(Ls, 1)).sum() == 0 and abs(sp.tril(Us, -1)).sum() == 0 and np.all(Ls.diagonal() == 1.0)
    assert r[0] == 0 and r[-1] == A.size[0]
    for M in (P, Q):                                                   # permutation matrices
        Ms = to_sp(M)
        assert np.all(Ms.sum(axis=0) == 1) and np.all(Ms.sum(axis=1) == 1)


@pytest.mark.parametrize("name", CASES)
def test_linsolve_and_solve(golden_dir, name):
    """test_sparse_solvers.py:238-284: A*x (or A'*x) reproduces b to 7 places, for linsolve and symbolic/numeric/solve."""
    A = load(golden_dir, name)
    As = to_sp(A)
    n = A.size[0]
    b = np.random.default_rng(3).standard_normal((n, 3))
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    for tran in "NT":
        M = As if tran == "N" else As.T
        x = matrix(b.copy())
        klu.linsolve(A, x, trans=tran)
        x1 = np.array(x._a).reshape(n, 3, order="F")
        assert np.abs(M @ x1 - b).max() < 5e-8
        y = matrix(b.copy())
        klu.solve(A, Fs, Fn, y, trans=tran)
        y1 = np.array(y._a).reshape(n, 3, order="F")
        assert np.abs(M @ y1 - b).max() < 5e-8
        assert np.linalg.norm(M @ y1 - b) <= 1e-10 * np.linalg.norm(b) * max(1.0, np.abs(y1).max())


@pytest.mark.parametrize("name", ["ACTIVSg2000", "bp_800"])
def test_parity_with_oracle(golden_dir, name):
    A = load(golden_dir, name)
    n = A.size[0]
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    e = Fn.num.extract()
    O = OracleKLU(n, A.colptr, A.rowind, A.values, Q=e["Q"])          # same column order: comparable fill, fast oracle
    b = np.random.default_rng(5).standard_normal((n, 2))
    for tran in "NT":
        x = np.asfortranarray(b.copy())
        klu.solve(A, Fs, Fn, x, trans=tran)
        xo = O.solve(b, tran)
        assert np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())


def test_block_triangular_form_of_bp_800(golden_dir):
    """KLU's R P A Q = L U + F with a non-trivial F: bp_800 has 492 diagonal blocks (471 singletons); L U is block
    diagonal, F strictly block upper triangular, r the block boundaries."""
    A = load(golden_dir, "bp_800")
    n = A.size[0]
    Fs = klu.symbolic(A)
    nb, nlev, blk = Fs.sym.btf()
    assert nb == 492 and 1 < nlev <= 64
    Fn = klu.numeric(A, Fs)
    L, U, P, Q, R, F, r = klu.get_numeric(A, Fs, Fn)
    assert len(r) == nb + 1 and r[0] == 0 and r[-1] == n and all(a < b for a, b in zip(r, r[1:]))
    bo = np.searchsorted(np.array(r), np.arange(n), side="right") - 1        # block of every pivotal position
    for Msp, name in ((to_sp(L), "L"), (to_sp(U), "U")):
        C = Msp.tocoo()
        assert np.all(bo[C.row] == bo[C.col]), name                           # factors of the diagonal blocks only
    C = to_sp(F).tocoo()
    assert C.nnz > 0 and np.all(bo[C.row] < bo[C.col])                        # F strictly above the block diagonal
    assert (to_sp(L).nnz - n) + to_sp(U).nnz + C.nnz < 1.6 * A.values.size    # 6 969 entries for 4 534 in A (9 574 + without BTF)


def test_doc_known_answers():
    A = spmatrix(DOC_V, DOC_I, DOC_J)
    B = matrix(np.arange(5.0))
    klu.linsolve(A, B)                                                 # spsolvers.rst:333-345
    assert np.allclose(B._a, [5.26e-02, -3.51e-02, 3.00e+00, 5.48e+00, -1.86e+00], rtol=5e-3)
    Bm = spmatrix(DOC_VB, DOC_I, DOC_J)                                # spsolvers.rst:420-439
    x = matrix(np.ones(5))
    Fs = klu.symbolic(A)
    FA = klu.numeric(A, Fs)
    FB = klu.numeric(Bm, Fs)
    klu.solve(A, Fs, FA, x)
    klu.solve(Bm, Fs, FB, x)
    klu.solve(A, Fs, FA, x, trans="T")
    assert np.allclose(x._a, [5.81e-01, -2.37e-01, 1.63e+00, 8.07e+00, -1.31e-01], rtol=5e-3)
    O = OracleKLU(5, A.colptr, A.rowind, A.values)
    OB = OracleKLU(5, Bm.colptr, Bm.rowind, Bm.values)
    assert np.allclose(x._a, O.solve(OB.solve(O.solve(np.ones(5))), "T"), rtol=1e-12)


def test_get_det():
    """test_sparse_solvers.py:286-310."""
    A = spmatrix(DOC_V, DOC_I, DOC_J)
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    det2 = np.linalg.det(to_sp(A).toarray())
    assert abs(klu.get_det(A, Fs, Fn) - det2) < 1e-7
    rng = np.random.default_rng(11)
    for n in (1, 2, 7, 40):
        D = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.5) + np.diag(rng.standard_normal(n))
        if abs(np.linalg.det(D)) < 1e-6:
            continue
        M = from_dense(D)
        Fs = klu.symbolic(M)
        Fn = klu.numeric(M, Fs)
        d = klu.get_det(M, Fs, Fn)
        assert abs(d - np.linalg.det(D)) <= 1e-9 * abs(np.linalg.det(D))
        assert abs(d - OracleKLU(n, M.colptr, M.rowind, M.values).det()) <= 1e-9 * abs(d)


def test_singular_matrices_raise_arithmetic_error():
    """klu.c:172-174, 370-371: ArithmeticError('singular matrix')."""
    M = from_dense(np.array([[1.0, 2.0, 0], [2.0, 4.0, 0], [0, 0, 1.0]]))                   # numerically singular
    Fs = klu.symbolic(M)
    with pytest.raises(ArithmeticError):
        klu.numeric(M, Fs)
    with pytest.raises(ArithmeticError):
        klu.linsolve(M, matrix(np.ones(3)))
    S = spmatrix([1.0, 1.0, 1.0, 1.0], [0, 1, 0, 1], [0, 0, 1, 1], (3, 3))                   # empty row and column
    with pytest.raises(ArithmeticError):
        klu.linsolve(S, matrix(np.ones(3)))
    # rank-deficient inside a larger sparse system: two identical rows far apart in the elimination order
    rng = np.random.default_rng(2)
    n = 60
    D = np.diag(2.0 + rng.random(n)) + (rng.random((n, n)) < 0.05) * rng.standard_normal((n, n))
    D[n - 1, :] = D[3, :]
    M = from_dense(D)
    with pytest.raises(ArithmeticError):
        klu.numeric(M, klu.symbolic(M))


def test_zero_diagonal_random_matrices_force_pivoting_and_merges():
    """Matrices with structurally zero diagonals and cancellations: every solution must match the oracle and dense LAPACK."""
    rng = np.random.default_rng(42)
    merges = 0
    for trial in range(40):
        n = int(rng.integers(5, 160))
        dens = rng.choice([0.03, 0.08, 0.2])
        D = (rng.random((n, n)) < dens) * rng.integers(-2, 3, (n, n)).astype(float)         # small integers: exact cancellations
        p = rng.permutation(n)
        D[p, np.arange(n)] += rng.choice([-1.0, 1.0, 2.0], n)                                # structurally nonsingular
        D[np.arange(n), np.arange(n)] *= (rng.random(n) < 0.5)                               # knock out half of the diagonal
        if np.linalg.matrix_rank(D) < n or np.linalg.cond(D) > 1e10:
            continue
        M = from_dense(D)
        Fs = klu.symbolic(M)
        Fn = klu.numeric(M, Fs)
        merges += Fs.sym.info()["merges"]
        b = rng.standard_normal((n, 2))
        for tran in "NT":
            x = np.asfortranarray(b.copy())
            klu.solve(M, Fs, Fn, x, trans=tran)
            ref = np.linalg.solve(D if tran == "N" else D.T, b)
            assert np.abs(x - ref).max() <= 1e-8 * max(1.0, np.abs(ref).max()), (trial, n, tran)
            xo = OracleKLU(n, M.colptr, M.rowind, M.values).solve(b, tran)
            assert np.abs(x - xo).max() <= 1e-8 * max(1.0, np.abs(xo).max())
        L, U, P, Q, R, F, r = klu.get_numeric(M, Fs, Fn)
        rho = abs(to_sp(R) @ to_sp(P) @ to_sp(M) @ to_sp(Q) - to_sp(L) @ to_sp(U) - to_sp(F)).sum(axis=0).max()
        assert rho < 1e-10
    assert merges > 0            # the merge path was exercised


def test_refactorisation_reuses_and_falls_back(golden_dir):
    A = load(golden_dir, "ACTIVSg2000")
    n = A.size[0]
    rng = np.random.default_rng(9)
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    p0 = Fn.num.info()["passes"]
    A2 = spmatrix.from_ccs(n, n, A.colptr, A.rowind, A.values * (1.0 + 0.05 * rng.random(A.values.size)))
    assert klu.numeric(A2, Fs, Fn) is Fn                               # spsolvers.rst:377-388
    assert Fn.num.info()["passes"] == p0 + 1                            # one pass, pivots reused
    b = rng.standard_normal(n)
    x = b.copy()
    klu.solve(A2, Fs, Fn, x)
    assert np.abs(to_sp(A2) @ x - b).max() < 5e-8
    # values that make the reused pivots unacceptable: the refactorisation becomes a full factorisation
    D = np.array([[4.0, 1.0, 0], [1.0, 3.0, 1.0], [0, 1.0, 2.0]])
    M = from_dense(D)
    Fs = klu.symbolic(M); Fn = klu.numeric(M, Fs)
    D2 = np.array([[1e-14, 1.0, 0], [1.0, 1e-14, 1.0], [0, 1.0, 1e-14]])
    M2 = from_dense(D2)
    klu.numeric(M2, Fs, Fn)
    x = np.ones(3); klu.solve(M2, Fs, Fn, x)
    assert np.allclose(D2 @ x, np.ones(3), atol=1e-12)


def test_unsymmetric_grid_uses_the_blocked_big_front_path():
    """Convection-diffusion on a 150 x 150 grid (n = 22 500): fronts of several hundred rows go through the blocked
    HBM path (panel / trsm / update); solution against SciPy's SuperLU and residual to the north-star bar."""
    import scipy.sparse.linalg as spla
    from kvxopt_amd import workloads
    n, cp, ri, v = workloads.convdiff_2d(150, seed=1)
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    As = to_sp(A)
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    assert Fn.num.info()["max_front"] > 136                              # beyond the LDS-resident class
    b = np.random.default_rng(8).standard_normal((n, 2))
    lu = spla.splu(As)
    for tran in "NT":
        x = np.asfortranarray(b.copy())
        klu.solve(A, Fs, Fn, x, trans=tran)
        M = As if tran == "N" else As.T
        assert np.linalg.norm(M @ x - b) <= 1e-10 * np.linalg.norm(b)
        xs = lu.solve(b, trans=tran)
        assert np.abs(x - xs).max() <= 1e-7 * max(1.0, np.abs(xs).max())
    L, U, P, Q, R, F, r = klu.get_numeric(A, Fs, Fn)
    rho = abs(to_sp(R) @ to_sp(P) @ As @ to_sp(Q) - to_sp(L) @ to_sp(U) - to_sp(F)).sum(axis=0).max()
    assert rho < 1e-12


def test_blocked_path_on_fronts_of_many_tiles_factor_and_refactor():
    """Convection-diffusion on a 300 x 300 grid: fronts of ~580 rows, i.e. nine tile rows per update launch (the U12 block is solved
    by every tile row of the launch and written once, never into the front the other rows read), 16- and 32-pivot panels.
    Factor, then refactor with other values on the recorded pivot sequence (klu.c:296-308): residual and L U = R P A Q - F."""
    from kvxopt_amd import workloads
    n, cp, ri, v = workloads.convdiff_2d(300, seed=2)
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    assert Fn.num.info()["max_front"] > 512                              # nine or more tile rows
    b = np.random.default_rng(9).standard_normal((n, 2))
    v2 = v * (1.0 + 0.25 * np.random.default_rng(10).standard_normal(v.size))
    for vals, fresh in ((v, True), (v2, False), (v, False)):
        Av = spmatrix.from_ccs(n, n, cp, ri, vals)
        As = to_sp(Av)
        if not fresh:
            klu.numeric(Av, Fs, Fn)                                      # refactorisation into the existing factor
        x = np.asfortranarray(b.copy())
        klu.solve(Av, Fs, Fn, x)
        assert np.linalg.norm(As @ x - b) <= 1e-10 * np.linalg.norm(b)
    L, U, P, Q, R, F, r = klu.get_numeric(A, Fs, Fn)
    As = to_sp(A)
    rho = abs(to_sp(R) @ to_sp(P) @ As @ to_sp(Q) - to_sp(L) @ to_sp(U) - to_sp(F)).sum(axis=0).max()
    assert rho < 1e-11
    # the steady state -- the same buffers, the recorded pivot sequence -- replays captured launch graphs (unless turned off)
    if os.environ.get("KVX_LU_GRAPH", "1") != "0":
        before = Fn.num.graph_replays()
        for _ in range(3):
            klu.numeric(A, Fs, Fn)
            x = np.asfortranarray(b.copy())
            klu.solve(A, Fs, Fn, x)
            assert np.linalg.norm(As @ x - b) <= 1e-10 * np.linalg.norm(b)
        assert Fn.num.graph_replays() >= before + 4


def test_ldb_offset_and_nrhs_arguments():
    """klu.c:619-628: nrhs / ldB / offsetB follow the BLAS conventions."""
    A = spmatrix(DOC_V, DOC_I, DOC_J)
    As = to_sp(A).toarray()
    Fs = klu.symbolic(A); Fn = klu.numeric(A, Fs)
    buf = np.full(2 + 7 * 3, -7.0)
    b = np.random.default_rng(1).standard_normal((5, 3))
    for c in range(3):
        buf[2 + 7 * c: 2 + 7 * c + 5] = b[:, c]
    klu.solve(A, Fs, Fn, buf, nrhs=2, ldB=7, offsetB=2)
    for c in range(2):
        assert np.allclose(As @ buf[2 + 7 * c: 2 + 7 * c + 5], b[:, c], atol=1e-12)
    assert np.array_equal(buf[2 + 14: 2 + 14 + 5], b[:, 2])            # third column untouched
    assert buf[0] == -7.0 and buf[7] == -7.0 and buf[8] == -7.0        # gaps untouched
    with pytest.raises(ValueError):
        klu.solve(A, Fs, Fn, buf, ldB=3)
    with pytest.raises(ValueError):
        klu.solve(A, Fs, Fn, buf, trans="X")
    with pytest.raises(TypeError):
        klu.solve(A, Fs, Fn, np.ones(4), ldB=5)                           # err_buf_len


def test_two_factors_of_one_symbolic_and_device_resident_solve():
    A = spmatrix(DOC_V, DOC_I, DOC_J); Bm = spmatrix(DOC_VB, DOC_I, DOC_J)
    Fs = klu.symbolic(A)
    FA, FB = klu.numeric(A, Fs), klu.numeric(Bm, Fs)
    b = np.arange(1.0, 6.0)
    buf = _lib.DeviceBuffer.from_array(b)
    FA.num.solve_dev(buf.ptr, "N", 1)
    FB.num.solve_dev(buf.ptr, "T", 1)
    x = buf.download(np.float64, 5)
    ref = np.linalg.solve(to_sp(Bm).toarray().T, np.linalg.solve(to_sp(A).toarray(), b))
    assert np.allclose(x, ref, rtol=1e-12)


def test_device_entry_points_are_ordered_behind_the_callers_kernels(golden_dir):
    """Stream contract (include/kvxhip.h): the *_dev entry points run on the library's own non-blocking streams and must order
    them behind what the caller already submitted to the null stream.  Here the values / the right-hand side are produced by
    kernels on the null stream IMMEDIATELY before the call, many times over a large buffer so that they are still running when
    the call is made; a missing ordering reads stale data."""
    A = load(golden_dir, "ACTIVSg2000")
    n = A.size[0]
    L = _lib.lib()
    Fs = klu.symbolic(A)
    Fn = klu.numeric(A, Fs)
    nnz = A.values.size
    big = _lib.DeviceBuffer.from_array(np.ones(1 << 24))              # keeps the null stream busy ahead of the small kernels
    vals = _lib.DeviceBuffer.from_array(A.values)
    b = np.random.default_rng(8).standard_normal(n)
    rhs = _lib.DeviceBuffer.from_array(np.zeros(n))
    src = _lib.DeviceBuffer.from_array(b)
    for scale in (2.0, 0.5, 4.0):
        for _ in range(20):
            _lib.raise_for(L.kvx_vec_scal_dev(1 << 24, 1.0000001, big.ptr))
        _lib.raise_for(L.kvx_vec_scal_dev(nnz, scale, vals.ptr))       # values := scale * values, then straight into the refactor
        Fn.num.refactor_dev(vals.ptr, nnz)
        for _ in range(20):
            _lib.raise_for(L.kvx_vec_scal_dev(1 << 24, 0.9999999, big.ptr))
        _lib.raise_for(L.kvx_vec_copy_dev(n, src.ptr, rhs.ptr))        # rhs := b on the null stream, then straight into the solve
        Fn.num.solve_dev(rhs.ptr, "N", 1)
        x = rhs.download(np.float64, n)
        cur = vals.download(np.float64, nnz)
        M = sp.csc_matrix((cur, A.rowind, A.colptr), shape=(n, n))
        assert np.abs(M @ x - b).max() < 5e-8, scale
        _lib.raise_for(L.kvx_vec_fill_dev(n, 0.0, rhs.ptr))


@pytest.mark.parametrize("name", CASES)
def test_complex_matrices_as_the_reference_tests_build_them(golden_dir, name):
    """test_sparse_solvers.py:86-95, 238-284 with `_complex = True`: A := A + A*1j, b := b*1j, trans in 'N', 'T', 'C'; the
    product with the solution reproduces b to 7 places.  Complex systems run through the real 2n x 2n embedding on the same
    kernels; get_numeric / get_det of a complex factor are refused."""
    A = load(golden_dir, name)
    n = A.size[0]
    Az = spmatrix.from_ccs(n, n, A.colptr, A.rowind, A.values * (1.0 + 1.0j))
    assert Az.typecode == "z"
    As = sp.csc_matrix((Az.values, Az.rowind, Az.colptr), shape=(n, n))
    b = np.random.default_rng(4).standard_normal((n, 2)) * 1j
    Fs = klu.symbolic(Az)
    Fn = klu.numeric(Az, Fs)
    for tran, M in (("N", As), ("T", As.T), ("C", As.conj().T)):
        x = matrix(b.copy())
        assert x.typecode == "z"
        klu.linsolve(Az, x, trans=tran)
        x1 = np.array(x._a).reshape(n, 2, order="F")
        assert np.abs(M @ x1 - b).max() < 5e-8, tran
        y = matrix(b.copy())
        klu.solve(Az, Fs, Fn, y, trans=tran)
        assert np.abs(M @ np.array(y._a).reshape(n, 2, order="F") - b).max() < 5e-8, tran
    Fn = klu.numeric(spmatrix.from_ccs(n, n, A.colptr, A.rowind, A.values * (2.0 - 0.5j)), Fs, Fn)   # refactorisation
    y = matrix(b.copy())
    klu.solve(Az, Fs, Fn, y)
    assert np.abs(As @ np.array(y._a).reshape(n, 2, order="F") * (2.0 - 0.5j) / (1.0 + 1.0j) - b).max() < 5e-8
    with pytest.raises(NotImplementedError):
        klu.get_numeric(Az, Fs, Fn)
    with pytest.raises(NotImplementedError):
        klu.get_det(Az, Fs, Fn)
    with pytest.raises(TypeError):
        klu.linsolve(Az, matrix(np.ones(n)))                           # real B with a complex A (klu.c:121-122)
    with pytest.raises(TypeError):
        klu.numeric(A, Fs)                                             # real A with the symbolic factor of a complex one


def test_linsolve_cache_hit_with_very_different_values(monkeypatch):
    """klu.linsolve keeps the analysis (value-aware matching, block triangular form, pivot sequence) of the last
    matrices by PATTERN.  A second matrix on the same pattern whose large entries sit elsewhere -- tiny diagonal, a dominant
    cyclic shift, explicit zeros where the first matching was -- must come out as accurately as from a cold call
    (KVX_LINSOLVE_CACHE=0 semantics: fresh analysis + factorisation)."""
    import scipy.sparse as sp
    from kvxopt_amd import klu
    from kvxopt_amd.base import matrix, spmatrix
    n = 400
    rng = np.random.default_rng(11)
    M = sp.random(n, n, 0.01, random_state=5, format="csc")
    shift = sp.csc_matrix((np.ones(n), (np.arange(n), (np.arange(n) + 1) % n)), shape=(n, n))
    P = ((M != 0) + sp.eye(n, format="csc") + shift).astype(float).tocsc(); P.sort_indices()      # the common pattern
    def with_values(diag, sh):
        V = P.copy()
        V.data = rng.uniform(-0.3, 0.3, V.nnz)
        V = V.tolil()
        for i in range(n):
            V[i, i] = diag[i]
            V[i, (i + 1) % n] = sh[i]
        V = V.tocsc(); V.sort_indices()
        return V
    A1 = with_values(np.full(n, 10.0), np.full(n, 0.1))                                     # diagonally dominant: matching = the diagonal
    A2 = with_values(np.where(np.arange(n) % 3 == 0, 0.0, 1e-9), np.full(n, 10.0))        # the shift dominates; zeros / dust on the old matching
    b = rng.standard_normal(n)
    def solve(V):
        # the values of V on the COMMON pattern P (explicit zeros stay stored: the cache keys on the pattern)
        Vc = sp.csc_matrix((np.asarray(V[P.nonzero()]).ravel(), P.nonzero()), shape=(n, n)); Vc.sort_indices()
        Vc = sp.csc_matrix((Vc.data, Vc.indices, Vc.indptr), shape=(n, n))                # explicit zeros stay stored
        A = spmatrix.from_ccs(n, n, Vc.indptr.astype(np.int64), Vc.indices.astype(np.int64), Vc.data.copy())
        x = matrix(b.copy())
        klu.linsolve(A, x)
        xv = np.asarray(x._a).reshape(-1)
        return float(np.abs(V @ xv - b).max() / (np.abs(V).sum(axis=1).max() * np.abs(xv).max() + np.abs(b).max()))
    klu.clear_cache()
    r1 = solve(A1)
    r2_hit = solve(A2)                       # cache hit on the pattern analysed with A1's values
    klu.clear_cache()
    r2_cold = solve(A2)
    assert r1 < 1e-13 and r2_cold < 1e-13
    assert r2_hit < 1e-12 and r2_hit <= 100 * max(r2_cold, 1e-16), (r2_hit, r2_cold)
