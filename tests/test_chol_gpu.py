"""GPU parity tests proper: the HIP supernodal Cholesky (through the C ABI) against the CPU oracle on
the same inputs, the reference's documented known answers, the reference's own test matrices, and
size-independent properties at BASELINE.json's full size.

Tolerance (BASELINE.json north_star): solution vectors within 1e-10 relative residual; factor
entries and solves are compared with the oracle to 1e-11 relative (both are plain FP64, only the
summation order differs)."""
import os

import numpy as np
import pytest

from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from oracle.kvx_oracle import OracleChol

pytestmark = pytest.mark.gpu

RTOL = 1e-11


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    _lib.require_device()      # fail loudly: these tests must never pass on a fallback


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def parity(n, cp, ri, vx, uplo="L", perm=None, opts=None, nrhs=3, seed=0, tol=RTOL):
    F = Factor(n, cp, ri, uplo, perm, opts)
    F.factorize(vx)
    O = OracleChol(n, cp, ri, uplo, F.perm())
    O.factorize(vx)
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, nrhs))
    for sys_ in range(9):
        X = np.asfortranarray(B.copy()); Xo = np.asfortranarray(B.copy())
        F.solve(X, sys=sys_)
        O.solve(Xo, sys=sys_)
        if sys_ in (6, 7, 8):
            assert np.array_equal(X, Xo), sys_       # permutations / identity: bit exact
        else:
            assert rel(X, Xo) < tol * 50, (sys_, rel(X, Xo))
    assert rel(F.diag(), O.diag()) < tol
    # factor entries: getfactor (supernodal pattern incl. explicit zeros) vs the oracle's L
    Lp, Li, Lx = F.get_factor()
    Op, Oi, Ox = O.L()
    dense_ok = n <= 400
    if dense_ok:
        A = np.zeros((n, n)); Bm = np.zeros((n, n))
        for j in range(n):
            A[Li[Lp[j]:Lp[j + 1]], j] = Lx[Lp[j]:Lp[j + 1]]
            Bm[Oi[Op[j]:Op[j + 1]], j] = Ox[Op[j]:Op[j + 1]]
        assert np.abs(A - Bm).max() < tol * np.abs(Bm).max()
    return F, O


def test_doc_known_answers():
    # doc/source/spsolvers.rst:555-563, 700-708 (linsolve / symbolic+numeric+solve), 759-765 (log det)
    cp = np.array([0, 2, 4, 5, 6]); ri = np.array([0, 2, 1, 3, 2, 3]); vx = np.array([10.0, 3, 5, -2, 5, 2])
    F = Factor(4, cp, ri)
    F.factorize(vx)
    X = np.asfortranarray(np.arange(8, dtype=float).reshape(4, 2, order="F"))
    F.solve(X)
    doc = np.array([[-1.46e-01, 4.88e-02], [1.33e+00, 4.00e+00], [4.88e-01, 1.17e+00], [2.83e+00, 7.50e+00]])
    assert np.allclose(X, doc, rtol=5e-3)
    assert abs(2.0 * np.sum(np.log(F.diag())) - 5.50533153593) < 1e-10
    Xp, Xi, Xx = F.spsolve(4, np.arange(5), np.arange(4), np.ones(4))          # splinsolve inverse :580-585
    inv = np.zeros((4, 4))
    for j in range(4):
        inv[Xi[Xp[j]:Xp[j + 1]], j] = Xx[Xp[j]:Xp[j + 1]]
    doc_inv = np.array([[1.22e-01, 0, -7.32e-02, 0], [0, 3.33e-01, 0, 3.33e-01],
                        [-7.32e-02, 0, 2.44e-01, 0], [0, 3.33e-01, 0, 8.33e-01]])
    assert np.allclose(inv, doc_inv, atol=6e-4)


@pytest.mark.parametrize("g", [1, 2, 5, 13, 40, 90])
def test_laplacian_parity(g):
    parity(*workloads.laplacian_2d(g), seed=g)


def test_other_patterns_parity():
    parity(*workloads.laplacian_2d(23, 57))
    parity(*workloads.laplacian_3d(11))
    parity(*workloads.stencil21_2d(31))


@pytest.mark.parametrize("seed,n,dens", [(1, 60, 0.2), (2, 300, 0.03), (3, 1500, 0.004), (4, 2500, 0.02)])
def test_random_spd_parity(seed, n, dens):
    """Random patterns: irregular fronts, big dense-ish fronts (MFMA trailing updates), many children."""
    import scipy.sparse as sp
    M = sp.random(n, n, dens, random_state=seed, format="csc")
    S = (M @ M.T + sp.eye(n) * (1.0 + seed)).tocsc()
    L = sp.tril(S).tocsc(); L.sort_indices()
    parity(n, L.indptr, L.indices, L.data, seed=seed)
    U = sp.triu(S).tocsc(); U.sort_indices()
    if n <= 300:
        parity(n, U.indptr, U.indices, U.data, uplo="U", seed=seed)


def test_user_perm_options_and_refactor():
    n, cp, ri, vx = workloads.laplacian_2d(33)
    rng = np.random.default_rng(5)
    parity(n, cp, ri, vx, perm=rng.permutation(n))
    parity(n, cp, ri, vx, opts={"ordering": 1})            # natural order: a deep, thin tree
    parity(n, cp, ri, vx, opts={"relax_small": 0, "relax_z1": 0.0, "relax_z2": 0.0, "relax_z3": 0.0, "leaf_cols": 0})
    parity(n, cp, ri, vx, opts={"nd_leaf": 6})
    # numeric refactor on the same symbolic (misc.py:1462): new values, same handle
    F, O = parity(n, cp, ri, vx)
    for it in range(3):
        v2 = vx * rng.uniform(0.9, 1.1) + 0.0
        v2[cp[:-1]] += rng.uniform(0.0, 1.0, n)             # keep it SPD (diagonal first in each column)
        F.factorize(v2)
        O.factorize(v2)
        b = rng.standard_normal(n); x = b.copy(); xo = b.copy()
        F.solve(x); O.solve(xo)
        assert rel(x, xo) < 1e-10


def test_other_triangle_ignored_ldb_offset():
    """cholmod.c:137-157 (other triangle ignored) and :467-472 (nrhs / ldB / offsetB)."""
    import scipy.sparse as sp
    n, cp, ri, vx = workloads.laplacian_2d(12)
    Lm = sp.csc_matrix((vx, ri, cp), shape=(n, n))
    full = (Lm + sp.triu(Lm.T * 7.0 + sp.eye(n) * 0, 1)).tocsc(); full.sort_indices()   # junk in the upper triangle
    F = Factor(n, full.indptr, full.indices, "L")
    F.factorize(full.data)
    rng = np.random.default_rng(1)
    ld, off, nrhs = n + 5, 3, 2
    buf = rng.standard_normal(off + ld * nrhs)
    keep = buf.copy()
    F.solve(buf, nrhs=nrhs, ldB=ld, offset=off)
    for r in range(nrhs):
        b = keep[off + r * ld: off + r * ld + n]
        x = buf[off + r * ld: off + r * ld + n]
        assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - b) / np.linalg.norm(b) < 1e-12
        pad = slice(off + r * ld + n, off + (r + 1) * ld)
        assert np.array_equal(buf[pad], keep[pad])           # padding rows untouched
    assert np.array_equal(buf[:off], keep[:off])
    with pytest.raises(ValueError):
        F.solve(buf, nrhs=1, ldB=n - 1)
    with pytest.raises(ValueError):
        F.solve(buf, sys=9)


def test_error_semantics():
    """ArithmeticError(minor) from numeric (documented, cholmod.c:308-310,376-379) and
    "singular matrix" from solve on a failed factor (:456); ValueError on a symbolic factor (:452)."""
    n, cp, ri, vx = workloads.laplacian_2d(20)
    F = Factor(n, cp, ri)
    with pytest.raises(ValueError):
        F.solve(np.ones(n))
    bad = vx.copy()
    bad[cp[150]] = -5.0
    O = OracleChol(n, cp, ri, "L", F.perm())
    with pytest.raises(ArithmeticError) as eo:
        O.factorize(bad)
    with pytest.raises(ArithmeticError) as e:
        F.factorize(bad)
    assert e.value.args[0] == eo.value.args[0]               # same failing column as the oracle
    with pytest.raises(ArithmeticError):
        F.solve(np.ones(n))
    with pytest.raises(ArithmeticError):
        F.diag()
    F.factorize(vx)                                          # the handle recovers
    x = np.ones(n); F.solve(x)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - 1) < 1e-10
    # NaN input is "not positive definite", not a hang
    bad = vx.copy(); bad[cp[7]] = np.nan
    with pytest.raises(ArithmeticError):
        F.factorize(bad)


@pytest.mark.parametrize("g,where", [(90, 0.02), (90, 0.6), (90, 0.97), (90, 0.999), (150, 0.995)])
def test_not_posdef_column_in_big_fronts(g, where):
    """The failing column reported from inside the blocked big-front path (16-column LDS blocks of the
    64-column panels, including the diagonal blocks factored by the trailing-update kernel) is the
    oracle's: `where` picks the pivot position in the PERMUTED order (late = top separators)."""
    n, cp, ri, vx = workloads.laplacian_2d(g)
    F = Factor(n, cp, ri)
    perm = F.perm()
    col = int(perm[min(n - 1, int(where * n))])
    bad = vx.copy()
    bad[cp[col]] = -1.0 if where < 0.9 else 1e-3               # indefinite leaf pivot / too-small separator pivot
    O = OracleChol(n, cp, ri, "L", perm)
    with pytest.raises(ArithmeticError) as eo:
        O.factorize(bad)
    with pytest.raises(ArithmeticError) as e:
        F.factorize(bad)
    assert e.value.args[0] == eo.value.args[0]
    F.factorize(vx)
    x = np.ones(n); F.solve(x)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - 1) < 1e-9 * np.sqrt(n)


def test_syrk128_variant_parity(monkeypatch):
    """The opt-in 128 x 128 LDS-staged trailing update (KVX_SYRK128_TILES) gives the same factor and
    solutions as the oracle, including the diagonal blocks it factors in its (0, 0) workgroup."""
    import scipy.sparse as sp
    monkeypatch.setenv("KVX_SYRK128_TILES", "1")
    monkeypatch.setenv("KVX_NO_GRAPH", "1")
    M = sp.random(2500, 2500, 0.02, random_state=4, format="csc")
    S = (M @ M.T + sp.eye(2500) * 5.0).tocsc()
    L = sp.tril(S).tocsc(); L.sort_indices()
    parity(2500, L.indptr, L.indices, L.data, seed=4)
    parity(*workloads.laplacian_2d(150), seed=150)
    # the two-level blocking of very large fronts, forced on here (per panel only the rest of an outer block of 1024 -- or
    # KVX_OUTER_BLOCK -- columns, then one rank-1024 update of the trailing matrix)
    monkeypatch.delenv("KVX_SYRK128_TILES")
    monkeypatch.setenv("KVX_TWO_LEVEL_M", "0")
    parity(2500, L.indptr, L.indices, L.data, seed=5)
    parity(*workloads.laplacian_2d(150), seed=151)
    monkeypatch.setenv("KVX_OUTER_BLOCK", "256")
    parity(2500, L.indptr, L.indices, L.data, seed=6)


def test_async_factor_then_solve_without_host_round_trip():
    """kvx_chol_factorize_async_dev followed by kvx_chol_solve_dev: the solve is queued behind the factorisation
    and the factor's status is examined afterwards -- same answers, and a non-positive pivot still raises."""
    from kvxopt_amd._lib import DeviceBuffer
    n, cp, ri, vx = workloads.laplacian_2d(70)
    F = Factor(n, cp, ri)
    b = np.random.default_rng(1).standard_normal(n)
    vd, xd = DeviceBuffer.from_array(vx), DeviceBuffer.from_array(b)
    F.factorize_dev(vd.ptr, sync=False)
    F.solve_dev(xd.ptr, 0, 1, n)
    x = xd.download(np.float64, n)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x.reshape(-1, 1)).ravel() - b) < 1e-10 * np.linalg.norm(b)
    bad = vx.copy(); bad[cp[int(F.perm()[n // 2])]] = -3.0
    F.factorize_dev(DeviceBuffer.from_array(bad).ptr, sync=False)
    with pytest.raises(ArithmeticError):
        F.solve_dev(xd.ptr, 0, 1, n)
    with pytest.raises(ArithmeticError):
        F.status()
    F.factorize_dev(vd.ptr, sync=False)                         # and the handle recovers
    xd2 = DeviceBuffer.from_array(b)
    F.solve_dev(xd2.ptr, 0, 1, n)
    assert np.allclose(xd2.download(np.float64, n), x, rtol=0, atol=0)


def test_empty_and_tiny():
    F = Factor(0, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64))
    F.factorize(np.zeros(0))
    F.solve(np.zeros(0))                                     # n == 0 early return (cholmod.c:468)
    F1 = Factor(1, [0, 1], [0])
    F1.factorize([4.0])
    x = np.array([2.0]); F1.solve(x)
    assert x[0] == 0.5 and F1.diag()[0] == 2.0
    # two live factors at once (Sf and Kf coexist in kkt_chol2, misc.py:1486)
    n, cp, ri, vx = workloads.laplacian_2d(9)
    A = Factor(n, cp, ri); B = Factor(n, cp, ri, opts={"ordering": 1})
    A.factorize(vx); B.factorize(2 * vx)
    xa = np.ones(n); xb = np.ones(n); A.solve(xa); B.solve(xb)
    assert np.allclose(xa, 2 * xb, rtol=1e-12)


@pytest.mark.parametrize("name", ["bcsstk13", "bcsstk24"])
def test_reference_test_matrices(golden_dir, name):
    """BASELINE.json configs[0]: linsolve on bcsstk13 (lower triangle as stored, uplo='L',
    B = default_rng(13).standard_normal((n,3))); checked against the oracle and by residual
    (cond 1.1e10: the normwise backward error is the meaningful figure)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    n, cp, ri, v = int(z["n"]), z["colptr"], z["rowind"], z["values"]
    B = np.random.default_rng(13).standard_normal((n, 3))
    F = Factor(n, cp, ri)
    F.factorize(v)
    X = np.asfortranarray(B.copy())
    F.solve(X)
    R = workloads.sym_matvec(n, cp, ri, v, X) - B
    diag = ri == np.repeat(np.arange(n), np.diff(cp))
    nrmA = np.sqrt(2 * np.sum(v ** 2) - np.sum(v[diag] ** 2))
    assert np.linalg.norm(R) / (nrmA * np.linalg.norm(X) + np.linalg.norm(B)) < 1e-14
    O = OracleChol(n, cp, ri, "L", F.perm())
    O.factorize(v)
    Xo = np.asfortranarray(B.copy()); O.solve(Xo)
    assert rel(F.diag(), O.diag()) < 1e-9                     # conditioning 1e10 amplifies rounding
    assert np.linalg.norm(X - Xo) / np.linalg.norm(Xo) < 1e-6


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (n = 1e6): properties that need no oracle run --
    residual <= 1e-10, linearity of the solve, sys 7/8 round trip, idempotent refactor."""
    n, cp, ri, vx = workloads.laplacian_2d(1000)
    F = Factor(n, cp, ri)
    F.factorize(vx)
    rng = np.random.default_rng(2)
    b = rng.standard_normal(n)
    x = b.copy(); F.solve(x)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - b) / np.linalg.norm(b) < 1e-10
    c = rng.standard_normal(n)
    X = np.asfortranarray(np.stack([c, 2.0 * b - 3.0 * c], axis=1))
    F.solve(X)
    assert rel(2.0 * x - 3.0 * X[:, 0], X[:, 1]) < 1e-9       # linearity
    y = b.copy(); F.solve(y, sys=7); F.solve(y, sys=8)
    assert np.array_equal(y, b)
    d1 = F.diag()
    F.factorize(vx)
    assert np.array_equal(d1, F.diag())                        # bitwise reproducible refactor
    x2 = b.copy(); F.solve(x2)
    assert np.array_equal(x, x2)


def test_config5_workload_properties_on_one_gpu():
    """BASELINE.json configs[4]'s workload -- the 7-point Laplacian on a cube -- through the HIP path at 100^3 (n = 1e6, an eighth
    of config 5's unknowns; top front of order ~14 000: the two-level blocked MFMA update path), by properties that need no
    oracle run: residual <= 1e-10, linearity, permutation round trip, bitwise reproducible refactor and solve, positive diagonal
    with the known log-determinant sign structure, and the failing column of a matrix made indefinite in its LAST column."""
    n, cp, ri, vx = workloads.laplacian_3d(100)
    F = Factor(n, cp, ri)
    assert F.info()["max_front"] >= 6144                       # (the threshold of the two-level update: this test must cross it)
    F.factorize(vx)
    rng = np.random.default_rng(5)
    b = rng.standard_normal(n)
    x = b.copy(); F.solve(x)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - b) / np.linalg.norm(b) < 1e-10
    c = rng.standard_normal(n)
    X = np.asfortranarray(np.stack([c, 2.0 * b - 3.0 * c], axis=1))
    F.solve(X)
    assert rel(2.0 * x - 3.0 * X[:, 0], X[:, 1]) < 1e-9
    y = b.copy(); F.solve(y, sys=7); F.solve(y, sys=8)
    assert np.array_equal(y, b)
    d1 = F.diag()
    assert np.all(d1 > 0) and np.all(d1 <= np.sqrt(6.0) + 1e-12)   # l_jj^2 <= a_jj = 6
    F.factorize(vx)
    assert np.array_equal(d1, F.diag())
    x2 = b.copy(); F.solve(x2)
    assert np.array_equal(x, x2)
    bad = vx.copy(); p = F.perm()
    bad[cp[int(p[n - 1])]] = -1.0                                  # the last pivot of the root front
    with pytest.raises(ArithmeticError) as e:
        F.factorize(bad)
    assert e.value.args[0] == n - 1


def test_config5_full_size_on_one_gpu():
    """BASELINE.json configs[4] AT ITS SIZE -- the 7-point Laplacian on the 200^3 grid, n = 8e6, 73 GB of factor -- on the one
    GPU of the box, by properties that need no oracle run (the CPU oracle would take an hour): residual <= 1e-10, linearity of
    the solve, permutation round trip, positive bounded diagonal, bitwise reproducible refactorisation and solve."""
    n, cp, ri, vx = workloads.laplacian_3d(200)
    assert n == 8000000
    F = Factor(n, cp, ri)
    inf = F.info()
    assert inf["max_front"] >= 40000 and inf["lsize"] * 8 > 50e9      # the root separator of the cube; tens of GB of panels
    F.factorize(vx)
    rng = np.random.default_rng(5)
    b = rng.standard_normal(n)
    x = b.copy(); F.solve(x)
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - b) / np.linalg.norm(b) < 1e-10
    c = rng.standard_normal(n)
    X = np.asfortranarray(np.stack([c, 2.0 * b - 3.0 * c], axis=1))
    F.solve(X)
    assert rel(2.0 * x - 3.0 * X[:, 0], X[:, 1]) < 1e-9
    y = b.copy(); F.solve(y, sys=7); F.solve(y, sys=8)
    assert np.array_equal(y, b)
    d1 = F.diag()
    assert np.all(d1 > 0) and np.all(d1 <= np.sqrt(6.0) + 1e-12)
    F.factorize(vx)
    assert np.array_equal(d1, F.diag())
    x2 = b.copy(); F.solve(x2)
    assert np.array_equal(x, x2)
    del F
    _lib.lib().kvx_dev_trim()                                          # hand the 100 GB back before the next test


def test_dbound_replaces_small_pivots():
    """cholmod.options['dbound'] (cholmod.c:116-117; CHOLMOD: diagonal entries of L below dbound are replaced by dbound),
    in every kernel class (wave, LDS, blocked), and the drop-the-row form used by the interior-point driver."""
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import spmatrix, matrix
    A = spmatrix([4.0, 1e-12, 9.0], [0, 1, 2], [0, 1, 2])
    cholmod.options["dbound"] = 1e-3
    try:
        F = cholmod.symbolic(A)
        cholmod.numeric(A, F)
        assert sorted(np.array(cholmod.diag(F)._a).ravel()) == pytest.approx([1e-3, 2.0, 3.0], rel=1e-12)
    finally:
        cholmod.options.clear()
    # a singular Laplacian (pure Neumann problem): the last pivot of every kernel class is ~ 0 or slightly negative
    for g in (6, 11, 30):                       # fronts of the wave, LDS and blocked classes
        n, cp, ri, vx = workloads.laplacian_2d(g)
        vx = vx.copy()
        # subtract the coupling from the diagonal: A = graph Laplacian, A 1 = 0
        import scipy.sparse as sp
        L = sp.csc_matrix((vx, ri, cp), shape=(n, n))
        Lf = L + sp.tril(L, -1).T
        off = np.asarray(abs(Lf - sp.diags(Lf.diagonal())).sum(axis=1)).ravel()
        Lg = sp.tril(Lf - sp.diags(Lf.diagonal()) + sp.diags(off)).tocsc(); Lg.sort_indices()
        with pytest.raises(ArithmeticError):
            F0 = Factor(n, Lg.indptr, Lg.indices); F0.factorize(Lg.data * 1.0 - 0.0)
            if F0.info()["minor"] == n:          # rounding may leave the last pivot tiny but positive
                raise ArithmeticError("positive by rounding")
        F1 = Factor(n, Lg.indptr, Lg.indices, opts={"dbound": 1e-6, "dbound_drop": 1})
        F1.factorize(Lg.data)                    # no exception: the pivot is replaced, its row drops out
        Afull = Lg + sp.tril(Lg, -1).T
        rng = np.random.default_rng(g)
        b = rng.standard_normal(n); b -= b.mean()            # consistent right-hand side
        x = b.copy(); F1.solve(x)
        assert np.abs(Afull @ x - b).max() < 1e-8 * max(1.0, np.abs(x).max())


def test_covariance_selection_example_drives_the_mirror():
    """The reference's own user script for this API, examples/doc/chap7/covsel.py (SURVEY 8(b) "what calls it"), restated with
    numpy for its dense algebra and kvxopt_amd.cholmod for everything it asks of CHOLMOD: symbolic once, numeric per Newton
    step and per line-search trial (ArithmeticError on a non-positive-definite trial, cholmod.c:308-310), diag for log det,
    solve with an n-column identity.  At the solution inv(K) agrees with Y on the sparsity pattern."""
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import spmatrix, matrix
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    n = 40
    Bm = sp.random(n, n, density=0.06, random_state=3, data_rvs=lambda k: rng.standard_normal(k))
    K0 = (Bm @ Bm.T + 4.0 * sp.identity(n)).tocsc()               # a sparse SPD "true" precision matrix
    Sig = np.linalg.inv(K0.toarray())
    pat = sp.tril(K0).tocoo()
    order = np.lexsort((pat.row, pat.col))
    I, J = pat.row[order], pat.col[order]
    Yv = Sig[I, J]                                                  # Y: sample covariance on the pattern of K
    m = len(I)
    D = np.nonzero(I == J)[0]
    Kv = np.where(I == J, 1.0, 0.0)
    K = spmatrix(Kv, I, J, (n, n))
    F = cholmod.symbolic(K)
    failures = 0
    for iters in range(100):
        cholmod.numeric(K, F)
        d = np.array(cholmod.diag(F)._a).ravel()
        Kinv = np.asfortranarray(np.eye(n))
        cholmod.solve(F, Kinv)
        grad = 2 * (Yv - Kinv[I, J])
        hess = 2 * (Kinv[np.ix_(I, J)] * Kinv[np.ix_(J, I)] + Kinv[np.ix_(I, I)] * Kinv[np.ix_(J, J)])
        v = np.linalg.solve(hess, -grad)
        sqntdecr = -grad @ v
        if sqntdecr < 1e-12:
            break
        dx = v.copy(); dx[D] *= 2
        f = -2.0 * np.log(d).sum()
        s_ = 1.0
        for _ in range(50):
            Kn = spmatrix(np.array(K.values) + s_ * dx, I, J, (n, n))
            try:
                cholmod.numeric(Kn, F)
            except ArithmeticError:
                failures += 1
                s_ *= 0.5
                continue
            dn = np.array(cholmod.diag(F)._a).ravel()
            fn = -2.0 * np.log(dn).sum() + 2 * s_ * (v @ Yv)
            if fn < f - 0.01 * s_ * sqntdecr:
                break
            s_ *= 0.5
        K = spmatrix(np.array(K.values) + s_ * dx, I, J, (n, n))
    assert iters < 40
    Kd = np.zeros((n, n)); Kd[I, J] = K.values; Kd = Kd + np.tril(Kd, -1).T
    assert np.abs(np.linalg.inv(Kd)[I, J] - Yv).max() < 1e-7       # optimality: inv(K) = Y on the pattern
    assert np.abs(Kd - K0.toarray()).max() < 1e-5                  # and here the maximiser is the true precision matrix


@pytest.mark.parametrize("g,h,nrhs", [(61, 47, 16), (61, 47, 18), (61, 47, 37), (300, 280, 18), (300, 280, 67)])
def test_many_right_hand_sides_blocked_kernels(g, h, nrhs):
    """From 16 right-hand sides on the leaf-subtree walks take blocks of four right-hand sides per wavefront
    (k_fwd_subtree_mr / k_bwd_subtree_mr) and, from 64 on, the big-front steps blocks of eight per workgroup (k_fwd_big_step_mr,
    k_bwd_big_init_mr, k_bwd_big_step_mr; the 300 x 280 grid has fronts with more than 256 pivot columns: several super-steps);
    18, 37 and 67 leave ragged last blocks.  Below 64 right-hand sides every column must equal the single-rhs solve of the same
    column bit for bit (same kernels, same operations in the same order); the blocks of eight sum the rows below a super-block
    in four column groups where the single-rhs step of a launch with few fronts uses sixteen (64-row workgroups): equal to
    1e-13 there.  All match the oracle."""
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    os.environ["KVX_WIDE_FROM"] = "0"            # (read at the device set-up of the factor: keeps the 67-rhs case on these kernels)
    try:
        F = Factor(n, cp, ri)
        F.factorize(v)
    finally:
        del os.environ["KVX_WIDE_FROM"]
    O = OracleChol(n, cp, ri, "L", F.perm())
    O.factorize(v)
    B = np.random.default_rng(nrhs).standard_normal((n, nrhs))
    for sys in (0, 4, 5):
        X = np.asfortranarray(B.copy())
        F.solve(X, sys=sys)
        Xo = np.asfortranarray(B.copy())
        O.solve(Xo, sys=sys)
        assert np.abs(X - Xo).max() / np.abs(Xo).max() < 1e-11
        for j in (0, nrhs // 2, nrhs - 1):
            xj = B[:, j].copy()
            F.solve(xj, sys=sys)
            if nrhs < 64:
                assert np.array_equal(xj, X[:, j]), (sys, j)
            else:
                assert np.abs(xj - X[:, j]).max() <= 1e-13 * np.abs(xj).max(), (sys, j)


@pytest.mark.parametrize("name,nrhs", [("grid61x47", 64), ("grid61x47", 70), ("grid300x280", 67), ("grid300x280", 130),
                                       ("cube24", 64), ("cube24", 200), ("random3000", 96)])
def test_many_right_hand_sides_rhs_major_blocks(name, nrhs):
    """From 64 right-hand sides on the solves run on rhs-major blocks of 64 (csrc/kernels_wide.hip: MFMA panel products, 16 x 16
    substitutions with one rhs per lane, extend-add through the inverse map; big fronts in several workgroups with one launch per
    64-column block).  70 / 67 / 130 / 200 leave ragged last blocks, the 300 x 280 grid and the cube have fronts with several
    64-column blocks and partial last ones.  A GEMM sums in another order than the substitution chains of the single-rhs kernels:
    the columns agree with the oracle and with single-rhs solves to 1e-12 (relative to the largest entry), not bit for bit; two
    calls (the second and third replay the captured graph) give identical bits."""
    if name.startswith("grid"):
        g, h = (int(t) for t in name[4:].split("x"))
        n, cp, ri, v = workloads.laplacian_2d(g, h)
    elif name.startswith("cube"):
        n, cp, ri, v = workloads.laplacian_3d(int(name[4:]))
    else:
        import scipy.sparse as sp
        n = int(name[6:])
        M = sp.random(n, n, 0.002, random_state=11, format="csc")
        L = sp.tril((M @ M.T + sp.eye(n) * 3.0).tocsc()).tocsc(); L.sort_indices()
        cp, ri, v = L.indptr, L.indices, L.data
    F = Factor(n, cp, ri)
    F.factorize(v)
    O = OracleChol(n, cp, ri, "L", F.perm())
    O.factorize(v)
    B = np.random.default_rng(nrhs).standard_normal((n, nrhs))
    for sys in (0, 4, 5):
        X = np.asfortranarray(B.copy())
        F.solve(X, sys=sys)
        Xo = np.asfortranarray(B.copy())
        O.solve(Xo, sys=sys)
        assert np.abs(X - Xo).max() / np.abs(Xo).max() < 1e-11, sys
        for j in (0, 63 % nrhs, nrhs // 2, nrhs - 1):
            xj = B[:, j].copy()
            F.solve(xj, sys=sys)
            assert np.abs(xj - X[:, j]).max() <= 1e-12 * np.abs(xj).max(), (sys, j)
        for _ in range(2):
            X2 = np.asfortranarray(B.copy())
            F.solve(X2, sys=sys)
            assert np.array_equal(X, X2), sys
    X = np.asfortranarray(B.copy())
    F.solve(X)
    R = workloads.sym_matvec(n, cp, ri, v, X) - B
    assert np.abs(R).max() / np.abs(B).max() < 1e-11


def test_device_pool_recycles_and_trims():
    """The caching pool behind every device buffer (csrc/devpool.cpp): a factor created after another one of the same shape
    was freed gets its blocks back (results unchanged), and kvx_dev_trim() returns the cached memory to the driver."""
    from kvxopt_amd._lib import lib, raise_for
    n, cp, ri, v = workloads.laplacian_2d(40, 33)
    b = np.random.default_rng(5).standard_normal(n)
    xs = []
    for rep in range(3):
        F = Factor(n, cp, ri)
        F.factorize(v * (1.0 + rep))
        x = b.copy()
        F.solve(x)
        xs.append(x * (1.0 + rep))
        del F
        if rep == 1:
            raise_for(lib().kvx_dev_trim())
    assert np.abs(xs[0] - xs[1]).max() < 1e-12 * np.abs(xs[0]).max() and np.abs(xs[0] - xs[2]).max() < 1e-12 * np.abs(xs[0]).max()
    r = workloads.sym_matvec(n, cp, ri, v, xs[0]) - b
    assert np.abs(r).max() < 1e-12 * np.abs(b).max()


def test_complex_hermitian_systems():
    """cholmod with 'z' matrices (cholmod.c:144,153,463): Hermitian positive definite A given by either triangle, complex
    right-hand sides -- symbolic / numeric / solve with every system code 0..8, linsolve, spsolve, splinsolve, diag and getfactor
    (cholmod.c:900-985 on a 'z' factor) against dense numpy: L = getfactor(F) is lower triangular with a real positive diagonal
    and P A P^T = L L^H for the P of sys = 7; the partial systems agree with that L.  Runs through the real 2n x 2n embedding
    in interleaved numbering, whose real factor is the embedding of the complex one (cholmod._embed_hermitian)."""
    import scipy.sparse as sp
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import matrix, spmatrix
    rng = np.random.default_rng(11)
    n = 60
    M = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * (rng.random((n, n)) < 0.08)
    Ad = M @ M.conj().T + np.diag(rng.uniform(2.0, 4.0, n))
    B = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    Xref = np.linalg.solve(Ad, B)
    for uplo, T in (("L", np.tril(Ad)), ("U", np.triu(Ad))):
        S = sp.csc_matrix(T); S.sort_indices()
        A = spmatrix.from_ccs(n, n, S.indptr, S.indices, S.data)
        assert A.typecode == "z"
        F = cholmod.symbolic(A, uplo=uplo)
        cholmod.numeric(A, F)
        X = matrix(B.copy(order="F"))
        cholmod.solve(F, X)
        assert rel(np.array(X.a), Xref) < 1e-11, uplo
        X2 = matrix(B.copy(order="F"))
        cholmod.linsolve(A, X2, uplo=uplo)
        assert rel(np.array(X2.a), Xref) < 1e-11
        cholmod.numeric(spmatrix.from_ccs(n, n, S.indptr, S.indices, 2.0 * S.data), F)           # refactor, new values
        X3 = matrix(B.copy(order="F")); cholmod.solve(F, X3)
        assert rel(np.array(X3.a), Xref / 2.0) < 1e-11
        Bs = spmatrix([1.0 + 2.0j, -1.0j, 3.0], [0, 7, 7], [0, 0, 1], (n, 2))
        Xs = cholmod.splinsolve(A, Bs, uplo=uplo)
        assert Xs.typecode == "z" and rel(Xs.todense(), np.linalg.solve(Ad, Bs.todense())) < 1e-11
        cholmod.numeric(A, F)
        Lm = cholmod.getfactor(F)
        assert Lm.typecode == "z"
        L = np.array(Lm.todense())
        assert np.allclose(np.triu(L, 1), 0.0) and np.all(np.diag(L).real > 0) and np.abs(np.diag(L).imag).max() == 0.0
        e = matrix(np.arange(n, dtype=np.complex128)); cholmod.solve(F, e, sys=7)          # x = P b: the permutation itself
        perm = np.array(e.a).real.astype(int).ravel()
        assert sorted(perm.tolist()) == list(range(n))
        PAPt = Ad[np.ix_(perm, perm)]
        assert rel(L @ L.conj().T, PAPt) < 1e-12
        assert rel(np.array(cholmod.diag(F).a).ravel(), np.diag(L)) < 1e-14
        P = np.eye(n)[perm]
        expect = {1: lambda b: np.linalg.solve(L @ L.conj().T, b), 2: lambda b: np.linalg.solve(L, b), 3: lambda b: np.linalg.solve(L.conj().T, b),
                  4: lambda b: np.linalg.solve(L, b), 5: lambda b: np.linalg.solve(L.conj().T, b), 6: lambda b: b,
                  7: lambda b: P @ b, 8: lambda b: P.T @ b}
        for sysc, fn in expect.items():
            Y = matrix(B.copy(order="F")); cholmod.solve(F, Y, sys=sysc)
            assert rel(np.array(Y.a), fn(B)) < 1e-11, (uplo, sysc)
        Ys = cholmod.spsolve(F, Bs, sys=4)
        assert Ys.typecode == "z" and rel(Ys.todense(), np.linalg.solve(L, np.array(Bs.todense()))) < 1e-11
        # ld / offset of a complex right-hand side
        buf = matrix(np.zeros((n + 5) * 3 + 2, dtype=np.complex128))
        arr = np.array(buf.a).reshape(-1)
        Bo = np.zeros((n + 5) * 3 + 2, dtype=np.complex128)
        for c in range(3):
            Bo[2 + c * (n + 5): 2 + c * (n + 5) + n] = B[:, c]
        bo = matrix(Bo.copy())
        cholmod.solve(F, bo, sys=0, nrhs=3, ldB=n + 5, offsetB=2)
        got = np.array(bo.a).reshape(-1)
        for c in range(3):
            assert rel(got[2 + c * (n + 5): 2 + c * (n + 5) + n], Xref[:, c]) < 1e-11
            assert np.all(got[2 + c * (n + 5) + n: 2 + (c + 1) * (n + 5)] == 0) if c < 2 else True
        with pytest.raises(TypeError):
            cholmod.solve(F, matrix(np.ones(n)))                       # real B with a complex factor (cholmod.c:461-465)
    bad = Ad.copy(); bad[5, 5] = -1.0
    S = sp.csc_matrix(np.tril(bad)); S.sort_indices()
    Ab = spmatrix.from_ccs(n, n, S.indptr, S.indices, S.data)
    with pytest.raises(ArithmeticError):
        cholmod.numeric(Ab, cholmod.symbolic(Ab))


@pytest.mark.gpu
@pytest.mark.parametrize("g,h,supernodal", [(160, 150, 2), (37, 41, 2), (60, 50, 0)])
def test_spsolve_forward_systems_sweep_only_the_reach(g, h, supernodal, monkeypatch):
    """spsolve with sys = 4 (L x = b) and 2 (L D x = b) on sparse right-hand sides (cholmod.c:524-587; what misc.kkt_chol2 uses to
    form L^-1 P A', misc.py:1483-1487) sweeps only the fronts that hold a nonzero row of a column block and their ancestors.
    The result must be what the dense column-block path gives (KVX_SPSOLVE_DENSE=1: every front swept) -- the same kernels on
    the same data: the same pattern bit for bit, the same values to rounding -- for 150 columns (three blocks, one ragged), empty columns, a duplicate
    entry and columns that reach the root through different subtrees; the 160 x 150 grid has big-class fronts."""
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    monkeypatch.setenv("KVX_WIDE_FROM", "0")     # (the dense reference path would otherwise take the rhs-major kernels: same values to rounding, not the same bits)
    F = Factor(n, cp, ri, opts={"supernodal": supernodal})
    F.factorize(v + 0.0)
    rng = np.random.default_rng(g)
    ncol = 150
    Bp, Bi, Bx = [0], [], []
    for j in range(ncol):
        cnt = 0 if j in (3, 77) else int(rng.integers(1, 4))
        rows = np.sort(rng.choice(n, size=cnt, replace=False))
        if j == 5 and cnt:
            rows = np.concatenate([rows, rows[:1]])              # a duplicate entry: summed
        Bi.extend(int(r) for r in rows)
        Bx.extend(rng.standard_normal(rows.size))
        Bp.append(len(Bi))
    Bp, Bi, Bx = np.array(Bp, dtype=np.int64), np.array(Bi, dtype=np.int64), np.array(Bx)
    for sys in (4, 2):
        monkeypatch.delenv("KVX_SPSOLVE_DENSE", raising=False)
        got = F.spsolve(ncol, Bp, Bi, Bx, sys=sys)
        monkeypatch.setenv("KVX_SPSOLVE_DENSE", "1")
        ref = F.spsolve(ncol, Bp, Bi, Bx, sys=sys)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), sys     # the patterns, bit for bit
        # values: the same kernels on the same data, except that the big-front forward step sums the rows below a super-block in
        # sixteen column groups (64-row workgroups, launches with few fronts, fewer than 64 right-hand sides) or in four (the
        # blocks of eight right-hand sides of the dense path): equal to rounding
        assert np.abs(got[2] - ref[2]).max() <= 1e-13 * np.abs(ref[2]).max(), sys
        assert got[0][-1] < 0.5 * n * ncol                       # L^-1 b is sparse here: the reach, not the whole vector
    # and against the definition, through a dense solve of one block
    monkeypatch.delenv("KVX_SPSOLVE_DENSE", raising=False)
    Xp, Xi, Xx = F.spsolve(ncol, Bp, Bi, Bx, sys=4)
    D = np.zeros((n, 8), order="F")
    for j in range(8):
        np.add.at(D[:, j], Bi[Bp[j]:Bp[j + 1]], Bx[Bp[j]:Bp[j + 1]])
    Y = D.copy(order="F")
    F.solve(Y, sys=4)
    for j in range(8):
        col = np.zeros(n)
        col[Xi[Xp[j]:Xp[j + 1]]] = Xx[Xp[j]:Xp[j + 1]]
        assert np.array_equal(col != 0, Y[:, j] != 0)                          # the same pattern ...
        assert np.abs(col - Y[:, j]).max() <= 1e-13 * np.abs(Y[:, j]).max()    # ... the same values to rounding (blocks of eight vs single columns)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["config2", "stencil21"])
def test_full_size_against_the_host_supernodal_oracle(which):
    """BASELINE.json configs[1] (5-point Laplacian 1000 x 1000, n = 1e6) and the north star's "~20 nnz/row" system (21-point
    stencil on the same grid) AT FULL SIZE against a CPU oracle: oracle/kvx_supernodal.c factors the same matrix on the same
    supernodes and permutation with OpenBLAS inside the fronts (different kernels, different summation order, the host's
    cores).  Solutions agree to 1e-10 relative (north_star)."""
    from oracle.kvx_oracle import OracleSupernodal
    n, cp, ri, vx = workloads.laplacian_2d(1000) if which == "config2" else workloads.stencil21_2d(1000)
    F = Factor(n, cp, ri)
    F.factorize(vx)
    b = np.random.default_rng(2).standard_normal(n)
    x = b.copy()
    F.solve(x)
    O = OracleSupernodal.from_factor(n, cp, ri, F)
    O.factorize(vx)
    xo = b.copy()
    O.solve(xo)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-10
    assert np.linalg.norm(workloads.sym_matvec(n, cp, ri, vx, x) - b) / np.linalg.norm(b) < 1e-10


@pytest.mark.gpu
def test_factor_subtree_walk_gives_the_same_factor(monkeypatch):
    """KVX_FACTOR_SUBTREES=1 (opt-in, measured slower): the leaf subtrees are factored by one wavefront each before the level
    loop, every front of a subtree in a slot of the update buffers of its own.  Same arithmetic per front: the factor and the
    solution are bit-identical to the level schedule's."""
    n, cp, ri, vx = workloads.laplacian_2d(220, 190)
    b = np.random.default_rng(5).standard_normal(n)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("KVX_FACTOR_SUBTREES", flag)
        F = Factor(n, cp, ri)
        F.factorize(vx)
        x = b.copy()
        F.solve(x)
        out.append((F.diag(), x))
        F.factorize(vx)                                     # graph replay from the second call on
        F.factorize(vx)
        assert np.array_equal(F.diag(), out[-1][0])
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("with_torch", ["0", "1"])
def test_graph_replay_with_the_hip_runtime_of_the_torch_wheel(with_torch):
    """A process that imports torch before this library runs on the HIP runtime bundled with the wheel (bench.py does).  Under that
    runtime the memset nodes of a captured factorisation were not ordered before the kernels behind them when a SMALL factor was
    replayed (the dense K of misc.kkt_chol2: `Terminated (singular KKT matrix)` from the second conelp call on); the prologue
    is kernels now.  A child process refactorises a sparse and a small dense factor alternately through their graphs and checks
    every solve, once per runtime."""
    import subprocess, sys
    env = dict(os.environ, WITH_TORCH=with_torch)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_replay_child.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "done, failures: 0" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_rhs_major_blocks_edge_cases():
    """The rhs-major path beyond its plain use: more right-hand sides than one pass of the workspace takes (1024: 1100 go in two
    passes), a natural ordering (a deep, thin tree: hundreds of levels, mostly chains of small fronts), a user permutation,
    a leading dimension larger than n; an LDL' view (options['supernodal'] = 0): the systems with D are the LL' sweeps with
    diag(Lc) applied in the layout kernels (sys 2 - 5), D x = b (sys 6) keeps the elementwise kernel."""
    n, cp, ri, v = workloads.laplacian_2d(30, 22)
    rng = np.random.default_rng(3)
    for opts, perm in (({}, None), ({"ordering": 1}, None), ({}, rng.permutation(n))):
        F = Factor(n, cp, ri, "L", perm, opts)
        F.factorize(v)
        O = OracleChol(n, cp, ri, "L", F.perm())
        O.factorize(v)
        nrhs = 1100 if not opts and perm is None else 80
        B = rng.standard_normal((n, nrhs))
        X = np.asfortranarray(B.copy())
        F.solve(X)
        Xo = np.asfortranarray(B.copy())
        O.solve(Xo)
        assert np.abs(X - Xo).max() / np.abs(Xo).max() < 1e-11
    # leading dimension > n
    ld = n + 7
    Bl = np.zeros((ld, 70), order="F")
    Bl[:n] = rng.standard_normal((n, 70))
    keep = Bl.copy()
    F.solve(Bl.reshape(-1, order="F"), nrhs=70, ldB=ld)
    Xo = np.asfortranarray(keep[:n].copy())
    O.solve(Xo)
    assert np.abs(Bl[:n] - Xo).max() / np.abs(Xo).max() < 1e-11 and np.array_equal(Bl[n:], keep[n:])
    # LDL' view
    Fl = Factor(n, cp, ri, opts={"supernodal": 0})
    Fl.factorize(v)
    B = rng.standard_normal((n, 66))
    for sys in (0, 1, 2, 3, 4, 5, 6):
        X = np.asfortranarray(B.copy())
        Fl.solve(X, sys=sys)
        for j in (0, 33, 65):
            xj = B[:, j].copy()
            Fl.solve(xj, sys=sys)
            if sys == 6:
                assert np.array_equal(xj, X[:, j]), (sys, j)
            else:
                assert np.abs(xj - X[:, j]).max() <= 1e-12 * np.abs(xj).max(), (sys, j)


@pytest.mark.parametrize("g,h,nrhs", [(90, 90, 1), (300, 280, 3), (33, 71, 2)])
def test_factor_and_solve_in_one_enqueue(g, h, nrhs):
    """kvx_chol_factorize_solve_dev: numeric factorisation + solve A X = B with the forward sweep pipelined behind the
    factorisation level by level (its own streams, one event per level).  Same kernels in the same order per front as
    factorize + solve: the solution must be bitwise that of the two separate calls -- eagerly (first call), from the captured
    graph (later calls), after a change of values, and the failing column of an indefinite matrix must come back."""
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    F = Factor(n, cp, ri)
    B = np.asfortranarray(np.random.default_rng(g + nrhs).standard_normal((n, nrhs)))
    Xref = B.copy(order="F")
    F.factorize(v)
    F.solve(Xref)
    vd = _lib.DeviceBuffer.from_array(v)
    bd = _lib.DeviceBuffer(8 * n * nrhs)
    for rep in range(4):                                        # eager, capture, replay, replay
        bd.upload(B.reshape(-1, order="F"))
        F.factorize_solve_dev(vd.ptr, bd.ptr, nrhs, n)
        X = bd.download(np.float64, n * nrhs).reshape(n, nrhs, order="F")
        assert np.array_equal(X, Xref), rep
    # the factor stays usable: a plain solve with it
    Y = B.copy(order="F"); F.solve(Y)
    assert np.array_equal(Y, Xref)
    # new values through the replayed graph
    v2 = v * 1.5
    vd.upload(v2)
    bd.upload(B.reshape(-1, order="F"))
    F.factorize_solve_dev(vd.ptr, bd.ptr, nrhs, n)
    X2 = bd.download(np.float64, n * nrhs).reshape(n, nrhs, order="F")
    assert np.abs(X2 - Xref / 1.5).max() <= 1e-12 * np.abs(Xref).max()
    # an indefinite matrix: the failing column, and the handle recovers
    bad = v.copy(); p = F.perm(); bad[cp[int(p[n // 2])]] = -1.0
    F2 = Factor(n, cp, ri)
    with pytest.raises(ArithmeticError) as e1:
        F2.factorize(bad)
    vd.upload(bad)
    bd.upload(B.reshape(-1, order="F"))
    with pytest.raises(ArithmeticError) as e2:
        F.factorize_solve_dev(vd.ptr, bd.ptr, nrhs, n)
    assert e1.value.args[0] == e2.value.args[0]
    vd.upload(v)
    bd.upload(B.reshape(-1, order="F"))
    F.factorize_solve_dev(vd.ptr, bd.ptr, nrhs, n)
    assert np.array_equal(bd.download(np.float64, n * nrhs).reshape(n, nrhs, order="F"), Xref)


@pytest.mark.gpu
def test_factorize_solve_with_host_buffers_and_linsolve():
    """kvx_chol_factorize_solve (host buffers: what cholmod.linsolve calls for a real LL' factor): the same answer as
    kvx_chol_factorize + kvx_chol_solve bit for bit, with ldB > n and an offset, repeated (cached symbolic factor, replayed graph)
    and with an indefinite matrix reported by its column; and linsolve against the CPU oracle."""
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import matrix, spmatrix
    n, cp, ri, v = workloads.laplacian_2d(37, 29)
    F = Factor(n, cp, ri)
    rng = np.random.default_rng(5)
    ld, off, nrhs = n + 3, 2, 3
    buf = rng.standard_normal(off + ld * nrhs)
    ref = buf.copy()
    F.factorize(v)
    F.solve(ref, 0, nrhs, ld, off)
    for rep in range(3):
        out = buf.copy()
        F.factorize_solve(v, out, nrhs=nrhs, ldB=ld, offset=off)
        assert np.array_equal(out, ref), rep                    # the gaps between the columns untouched as well
    bad = v.copy(); bad[cp[int(F.perm()[n // 3])]] = -2.0
    with pytest.raises(ArithmeticError) as e1:
        Factor(n, cp, ri).factorize(bad)
    with pytest.raises(ArithmeticError) as e2:
        F.factorize_solve(bad, buf.copy(), nrhs=nrhs, ldB=ld, offset=off)
    assert e1.value.args[0] == e2.value.args[0]
    # through the reference-facing call
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    cholmod.clear_cache()
    for rep in range(3):                                         # analysis + eager, capture, replay
        B = matrix(buf[off:off + ld * nrhs].reshape(ld, nrhs, order="F").copy(order="F"))
        cholmod.linsolve(A, B, nrhs=nrhs, ldB=ld)
        got = np.array(B.a).reshape(ld, nrhs, order="F")
        assert np.array_equal(got, ref[off:].reshape(ld, nrhs, order="F")), rep
    with pytest.raises(ArithmeticError):
        cholmod.linsolve(spmatrix.from_ccs(n, n, cp, ri, bad), matrix(buf[:n].copy()))


@pytest.mark.gpu
def test_fused_first_diagonal_block_is_bitwise():
    """k_assemble_big_potrf (round 3): at the top of the tree the first 64 x 64 diagonal block of a big front is assembled in LDS --
    A's entries, then the children in the order the extend-add takes them -- and factored by a workgroup of the extend-add launch.
    The factor and a solve are bit for bit those of the separate k_potrf_blk launch (KVX_ASM_POTRF_WGS=0)."""
    import json
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "factor_bits_probe.py")
    outs = []
    for flag in ("0", "1024"):
        env = dict(os.environ, KVX_ASM_POTRF_WGS=flag)
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
    assert all(v["max_front"] > 128 for v in outs[0].values())      # every system has big fronts (the path under test)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["no_graph", "torch_first"])
def test_two_enqueue_form_reports_like_the_one_enqueue_form(mode):
    """The fall-back of kvx_chol_factorize_solve* (factorisation and solve as two enqueues: KVX_NO_GRAPH=1, and every process that
    imported torch first and therefore runs on the HIP runtime of the wheel, older than 7.2) gives bit for bit the answer of the
    two calls and reports an indefinite matrix as the one-enqueue form and numeric() do: ArithmeticError(failing column),
    cholmod.c:308-310 -- not solve's 'singular matrix' (round-3 advisor finding).  A process of its own: the runtime is chosen by
    what a process loads first."""
    import json, os, subprocess, sys
    env = dict(os.environ)
    env.pop("KVX_NO_GRAPH", None)
    if mode == "no_graph":
        env["KVX_NO_GRAPH"] = "1"
    else:
        pytest.importorskip("torch")
        env["WITH_TORCH"] = "1"
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fused_path_child.py")
    r = subprocess.run([sys.executable, child], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["equal"]
    assert out["paths"] == [2, 2, 2], out                   # the fall-back was taken (on the 7.2 runtime without KVX_NO_GRAPH: [1, 1, 1])
    assert out["minor_fused"] == out["minor_numeric"] and out["minor_linsolve"] == out["minor_numeric"], out


@pytest.mark.gpu
def test_solution_bits_do_not_depend_on_nrhs(monkeypatch):
    """With the rhs-major path off (KVX_WIDE_FROM=0: the column-major kernels at any count), a right-hand side solved alone and
    as one of 64 has the same bits: the rhs-blocked big-front steps keep to the row association of the single-rhs sweep, also on
    the levels where that sweep takes 64-row workgroups (round-3 advisor finding).  150 x 140 grid: big fronts on several levels,
    narrow ones at the top."""
    monkeypatch.setenv("KVX_WIDE_FROM", "0")
    n, cp, ri, v = workloads.laplacian_2d(150, 140)
    F = Factor(n, cp, ri)
    F.factorize(v)
    rng = np.random.default_rng(64)
    B = np.asfortranarray(rng.standard_normal((n, 64)))
    X64 = B.copy(order="F")
    F.solve(X64.reshape(-1, order="F"), 0, 64, n)
    X64 = X64 if X64.flags.f_contiguous else np.asfortranarray(X64)
    for j in (0, 17, 63):
        x1 = B[:, j].copy()
        F.solve(x1, 0, 1, n)
        assert np.array_equal(x1, X64[:, j]), j
    X3 = np.asfortranarray(B[:, :3].copy())
    F.solve(X3.reshape(-1, order="F"), 0, 3, n)
    assert np.array_equal(X3, X64[:, :3])
