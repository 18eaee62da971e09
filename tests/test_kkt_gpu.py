"""GPU parity of the KKT-step pieces around the Cholesky factor -- NT scaling, normal-equations
assembly, sparse mat-vec, misc.kkt_chol2 and the device-resident LP driver -- against golden vectors
generated from the REFERENCE (tests/golden/make_goldens.py) and against the CPU oracle.

Floating-point tolerance: elementwise kernels are compared to 1e-14 relative (identical formulas,
one rounding each); assembled sums / solves to 1e-10 relative (BASELINE.json north_star bar)."""
import array
import json
import os

import numpy as np
import pytest

from kvxopt_amd import _lib, base, cholmod, lp, misc, workloads
from kvxopt_amd.base import matrix, spmatrix
from oracle import kvx_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    _lib.require_device()


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) if a.size else 0.0


def W_of(d, di):
    return {"d": matrix(d), "di": matrix(di), "v": [], "beta": [], "r": [], "rti": []}


@pytest.mark.parametrize("ml", [1, 7, 1000])
def test_nt_scaling_golden(golden_dir, ml):
    """G1: misc.compute_scaling / update_scaling / ssqr and misc_solvers.scale / scale2 / sprod / sinv /
    sdot / max_step of the reference on seeded inputs."""
    g = np.load(os.path.join(golden_dir, "g1_nt_scaling.npz"))
    k = lambda name: g["ml%d_%s" % (ml, name)]
    dims = {"l": ml, "q": [], "s": []}
    lm = matrix(0.0, (ml, 1))
    W = misc.compute_scaling(matrix(k("s")), matrix(k("z")), lm, dims)
    assert rel(W["d"]._a, k("d")) < 1e-14 and rel(W["di"]._a, k("di")) < 1e-14 and rel(lm._a, k("lmbda")) < 1e-14
    for tr in "NT":
        for inv in "NI":
            x = matrix(k("X").copy(order="F"))
            misc.scale(x, W, trans=tr, inverse=inv)
            assert rel(x.a, k("scale_%s%s" % (tr, inv))) < 1e-14
    x1, y1 = k("x1"), k("y1")
    a = matrix(x1.copy()); misc.scale2(lm, a, dims); assert rel(a._a, k("scale2_N")) < 1e-14
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, inverse="I"); assert rel(a._a, k("scale2_I")) < 1e-14
    a = matrix(x1.copy()); misc.sprod(a, matrix(y1), dims); assert rel(a._a, k("sprod")) < 1e-14
    a = matrix(x1.copy()); misc.sinv(a, matrix(y1), dims); assert rel(a._a, k("sinv")) < 1e-14
    a = matrix(0.0, (ml, 1)); misc.ssqr(a, matrix(x1), dims); assert rel(a._a, k("ssqr")) < 1e-14
    assert abs(misc.sdot(matrix(x1), matrix(y1), dims) - float(k("sdot"))) <= 1e-12 * max(1.0, abs(float(k("sdot"))))
    assert misc.max_step(matrix(x1), dims) == float(k("max_step"))
    W2 = W_of(k("d").copy(), k("di").copy())
    lm2, ms, mz = matrix(k("lmbda").copy()), matrix(k("us_ds").copy()), matrix(k("us_dz").copy())
    misc.update_scaling(W2, lm2, ms, mz)
    for got, name in ((ms, "us_s"), (mz, "us_z"), (W2["d"], "us_d"), (W2["di"], "us_di"), (lm2, "us_lmbda")):
        assert rel(got._a, k(name)) < 1e-14, name


def test_assembly_golden(golden_dir):
    """G2: base.gemm(spdiag(di), G, Gs, partial=True), base.syrk full + partial, S += H."""
    g = np.load(os.path.join(golden_dir, "g2_assembly.npz"))
    G = spmatrix.from_ccs(40, 12, g["G_cp"], g["G_ri"], g["G_v"])
    Gs = spmatrix.from_ccs(40, 12, g["G_cp"].copy(), g["G_ri"].copy(), np.zeros(g["G_v"].size))
    base.gemm(base.spdiag(matrix(g["di"])), G, Gs, partial=True)
    assert rel(Gs.values, g["Gs_v"]) < 1e-15
    S = spmatrix([], [], [], (12, 12))
    base.syrk(Gs, S, trans="T")
    assert np.array_equal(S.colptr, g["S_cp"]) and np.array_equal(S.rowind, g["S_ri"])      # same pattern as the reference
    assert rel(S.values, g["S_v"]) < 1e-13
    base.gemm(base.spdiag(matrix(g["di2"])), G, Gs, partial=True)
    base.syrk(Gs, S, trans="T", partial=True)
    assert rel(S.values, g["S2_v"]) < 1e-13
    S += spmatrix.from_ccs(12, 12, g["H_cp"], g["H_ri"], g["H_v"])
    assert np.array_equal(S.colptr, g["SH_cp"]) and np.array_equal(S.rowind, g["SH_ri"]) and rel(S.values, g["SH_v"]) < 1e-13


@pytest.mark.parametrize("tag,p", [("p0", 0), ("p3", 3)])
def test_kkt_chol2_golden(golden_dir, tag, p):
    """G3: misc.kkt_chol2(G, dims, A): first call with W = I (fixes the pattern), second call with the
    real scaling (numeric refactor on the same symbolic), then one solve -- against the reference's
    dense-G LAPACK branch (pure reference) on the same inputs.  p = 0 exercises the 0 x 0 K path."""
    g = np.load(os.path.join(golden_dir, "g3_kkt_chol2.npz"))
    ml, n = 30, 8
    G = spmatrix.from_ccs(ml, n, g[tag + "_G_cp"], g[tag + "_G_ri"], g[tag + "_G_v"])
    A = spmatrix.from_ccs(p, n, g[tag + "_A_cp"], g[tag + "_A_ri"], g[tag + "_A_v"]) if p else spmatrix([], [], [], (0, n))
    dims = {"l": ml, "q": [], "s": []}
    f = misc.kkt_chol2(G, dims, A)
    f(W_of(np.ones(ml), np.ones(ml)))
    solve = f(W_of(g[tag + "_d"].copy(), 1.0 / g[tag + "_d"]))
    x, y, z = matrix(g[tag + "_bx"].copy()), matrix(g[tag + "_by"].copy(), (p, 1)), matrix(g[tag + "_bz"].copy())
    solve(x, y, z)
    assert rel(x._a, g[tag + "_x"]) < 1e-10 and rel(z._a, g[tag + "_z"]) < 1e-10
    if p:
        assert rel(y._a, g[tag + "_y"]) < 1e-10
        # the mixed branch (sparse G, DENSE A; misc.py:1476-1481) gives the same triple
        f = misc.kkt_chol2(G, dims, matrix(A.todense()))
        f(W_of(np.ones(ml), np.ones(ml)))
        solve = f(W_of(g[tag + "_d"].copy(), 1.0 / g[tag + "_d"]))
        x, y, z = matrix(g[tag + "_bx"].copy()), matrix(g[tag + "_by"].copy(), (p, 1)), matrix(g[tag + "_bz"].copy())
        solve(x, y, z)
        assert rel(x._a, g[tag + "_x"]) < 1e-10 and rel(y._a, g[tag + "_y"]) < 1e-10 and rel(z._a, g[tag + "_z"]) < 1e-10
    # a DENSE G (the branch the golden itself was produced with in the reference, misc.py:1401-1404, 1464-1472) and a dense A
    f = misc.kkt_chol2(matrix(G.todense()), dims, matrix(A.todense()) if p else A)
    f(W_of(np.ones(ml), np.ones(ml)))
    solve = f(W_of(g[tag + "_d"].copy(), 1.0 / g[tag + "_d"]))
    x, y, z = matrix(g[tag + "_bx"].copy()), matrix(g[tag + "_by"].copy(), (p, 1)), matrix(g[tag + "_bz"].copy())
    solve(x, y, z)
    assert rel(x._a, g[tag + "_x"]) < 1e-10 and rel(z._a, g[tag + "_z"]) < 1e-10
    if p:
        assert rel(y._a, g[tag + "_y"]) < 1e-10


def test_nonlinear_block_scaling_golden(golden_dir):
    """G12: misc.compute_scaling / scale / update_scaling with the nonlinear block of cvxprog (mnl leading entries,
    W['dnl'], W['dnli']; misc.py:262-270, 48-60, 432-442) against the reference."""
    g = np.load(os.path.join(golden_dir, "g12_nonlinear_block.npz"))
    mnl, ml = 5, 30
    dims = {"l": ml, "q": [], "s": []}
    lm = matrix(0.0, (mnl + ml, 1))
    W = misc.compute_scaling(matrix(g["s"]), matrix(g["z"]), lm, dims, mnl)
    for k in ("dnl", "dnli", "d", "di"):
        assert W[k].size == (g["cs_" + k].size, 1) and rel(W[k]._a, g["cs_" + k]) < 1e-14, k
    assert rel(lm._a, g["cs_lmbda"]) < 1e-14
    for inv in "NI":
        x = matrix(g["scale_in"].copy(order="F"))
        misc.scale(x, W, trans="T", inverse=inv)
        assert rel(x.a, g["scale_" + inv]) < 1e-14
    ms, mz = matrix(g["us_s_in"].copy()), matrix(g["us_z_in"].copy())
    misc.update_scaling(W, lm, ms, mz)
    for got, name in ((ms, "us_s"), (mz, "us_z"), (lm, "us_lmbda"), (W["dnl"], "us_dnl"), (W["dnli"], "us_dnli"),
                      (W["d"], "us_d"), (W["di"], "us_di")):
        assert rel(got._a, g[name]) < 1e-14, name
    # mnl = 0 is not mnl = None: cvxprog passes 0 for problems without nonlinear constraints and still expects the keys
    W0 = misc.compute_scaling(matrix(g["s"][:ml]), matrix(g["z"][:ml]), matrix(0.0, (ml, 1)), dims, 0)
    assert W0["dnl"].size == (0, 1) and W0["d"].size == (ml, 1)


@pytest.mark.parametrize("tag,p", [("p0", 0), ("p2", 2)])
def test_kkt_chol2_nonlinear_block_golden(golden_dir, tag, p):
    """G12: misc.kkt_chol2(G, dims, A, mnl)(W, H, Df) -- the cvxprog form of the path (misc.py:1396-1400, 1413-1415,
    1423-1424, 1452-1453, 1523, 1560-1561): S = Df' Wnl^-2 Df + G' Wl^-2 G + H.  First call fixes the patterns, the
    second refactors with new W, H, Df values; the result is compared with the reference's dense LAPACK branch."""
    g = np.load(os.path.join(golden_dir, "g12_nonlinear_block.npz"))
    mnl, ml, n = 5, 30, 9
    G = spmatrix.from_ccs(ml, n, g[tag + "_G_cp"], g[tag + "_G_ri"], g[tag + "_G_v"])
    A = spmatrix.from_ccs(p, n, g[tag + "_A_cp"], g[tag + "_A_ri"], g[tag + "_A_v"]) if p else spmatrix([], [], [], (0, n))
    Df = [spmatrix.from_ccs(mnl, n, g[tag + "_Df_cp"], g[tag + "_Df_ri"], g[tag + "_Df%d_v" % i]) for i in (1, 2)]
    H = [spmatrix.from_ccs(n, n, g[tag + "_H_cp"], g[tag + "_H_ri"], g[tag + "_H%d_v" % i]) for i in (1, 2)]
    Ws = []
    for i in range(2):
        W = W_of(g["%s_W%d_d" % (tag, i)].copy(), 1.0 / g["%s_W%d_d" % (tag, i)])
        W["dnl"], W["dnli"] = matrix(g["%s_W%d_dnl" % (tag, i)].copy()), matrix(1.0 / g["%s_W%d_dnl" % (tag, i)])
        Ws.append(W)
    f = misc.kkt_chol2(G, {"l": ml, "q": [], "s": []}, A, mnl)
    f(Ws[0], H[0], Df[0])
    solve = f(Ws[1], H[1], Df[1])
    x, y, z = matrix(g[tag + "_bx"].copy()), matrix(g[tag + "_by"].copy(), (p, 1)), matrix(g[tag + "_bz"].copy())
    solve(x, y, z)
    assert rel(x._a, g[tag + "_x"]) < 1e-10 and rel(z._a, g[tag + "_z"]) < 1e-10
    if p:
        assert rel(y._a, g[tag + "_y"]) < 1e-10
    # dense Df and H (the reference's LAPACK branches): every entry stored, the same kernels, the same triple
    def sym_full(Hs):
        D = Hs.todense()
        return matrix(D + np.tril(D, -1).T)
    f = misc.kkt_chol2(G, {"l": ml, "q": [], "s": []}, A, mnl)
    f(Ws[0], sym_full(H[0]), matrix(Df[0].todense()))
    solve = f(Ws[1], sym_full(H[1]), matrix(Df[1].todense()))
    x, y, z = matrix(g[tag + "_bx"].copy()), matrix(g[tag + "_by"].copy(), (p, 1)), matrix(g[tag + "_bz"].copy())
    solve(x, y, z)
    assert rel(x._a, g[tag + "_x"]) < 1e-10 and rel(z._a, g[tag + "_z"]) < 1e-10
    if p:
        assert rel(y._a, g[tag + "_y"]) < 1e-10


def test_kkt_chol2_rejects_other_cones():
    with pytest.raises(ValueError):
        misc.kkt_chol2(spmatrix([1.0], [0], [0], (3, 1)), {"l": 0, "q": [3], "s": []}, spmatrix([], [], [], (0, 1)))


def test_atda_and_spmv_vs_oracle():
    """Bigger assembly / mat-vec against the CPU oracle (row scale + partial syrk of sparse.c)."""
    P = workloads.lp_grid(40, 30)
    ml, n = P["ml"], P["n"]
    rng = np.random.default_rng(9)
    di = rng.uniform(0.3, 3.0, ml)
    kkt = lp.KKTChol2Dev(ml, n, P["Gp"], P["Gi"], P["Gx"])
    dv = lp.DVec(ml, di)
    kkt.factor(dv)
    Sx = kkt.Sx.get()
    Gs, Sref = orc.atda(ml, n, P["Gp"], P["Gi"], P["Gx"], di, kkt.Sp, kkt.Si)
    assert rel(Sx, Sref) < 1e-13
    x = rng.standard_normal(n); y = rng.standard_normal(ml)
    yd, xd = lp.DVec(ml, y), lp.DVec(n, x)
    kkt.G.gemv(xd, yd, trans="N", alpha=2.0, beta=-0.5)
    yo = y.copy(); orc.spmv("N", ml, n, P["Gp"], P["Gi"], P["Gx"], x, yo, 2.0, -0.5)
    assert rel(yd.get(), yo) < 1e-13
    kkt.G.gemv(yd, xd, trans="T", alpha=-1.0, beta=1.0)
    xo = x.copy(); orc.spmv("T", ml, n, P["Gp"], P["Gi"], P["Gx"], yo, xo, -1.0, 1.0)
    assert rel(xd.get(), xo) < 1e-12
    # KKT solve against a dense solve of the full KKT system
    bx, bz = rng.standard_normal(n), rng.standard_normal(ml)
    xv, zv = lp.DVec(n, bx), lp.DVec(ml, bz)
    kkt.solve(xv, zv)
    Gd = np.zeros((ml, n)); Gd[P["Gi"], np.repeat(np.arange(n), np.diff(P["Gp"]))] = P["Gx"]
    d = 1.0 / di
    K = np.block([[np.zeros((n, n)), Gd.T], [Gd, -np.diag(d * d)]])
    sol = np.linalg.solve(K, np.concatenate([bx, bz]))
    assert rel(xv.get(), sol[:n]) < 1e-9 and rel(zv.get(), d * sol[n:]) < 1e-9


@pytest.mark.parametrize("name,gx,gy", [("grid6x5", 6, 5), ("grid25x20", 25, 20)])
def test_conelp_golden(golden_dir, name, gx, gy):
    """G4: the reference's conelp on the structured grid LP (config 4b generator): same iteration
    count, same NT scaling at every iteration, same solution."""
    g = np.load(os.path.join(golden_dir, "g4_conelp.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_conelp.json")))["cases"][name]
    P = workloads.lp_grid(gx, gy)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    sol = lp.conelp(P["c"], G, P["h"])
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    assert rel(sol["x"], g[name + "_x"]) < 1e-7 and rel(sol["s"], g[name + "_s"]) < 1e-6 and rel(sol["z"], g[name + "_z"]) < 1e-6
    assert abs(sol["primal objective"] - meta["primal objective"]) < 1e-7 * max(1.0, abs(meta["primal objective"]))
    assert sol["factorizations"] == meta["iterations"] + 1           # 1 initial + 1 per iteration, symbolic reused


def test_conelp_doc_and_infeasible(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "g4_conelp.json")))["cases"]
    # examples/doc/chap8/lp.py (tests/test_examples.py:31-34 expects x = [1, 1] to 5 places)
    G = spmatrix([2., 1., -1., 1., 2., -1.], [0, 1, 2, 0, 1, 3], [0, 0, 0, 1, 1, 1], (4, 2))
    sol = lp.conelp(np.array([-4., -5.]), G, np.array([3., 3., 0., 0.]))
    assert sol["status"] == "optimal" and sol["iterations"] == meta["doc_lp"]["iterations"]
    assert np.allclose(sol["x"], [1.0, 1.0], atol=1e-5) and np.allclose(sol["x"], meta["doc_lp"]["x"], atol=1e-8)
    G = spmatrix([-1.0, 1.0], [0, 1], [0, 0], (2, 1))
    sol = lp.conelp(np.array([1.0]), G, np.array([-1.0, 0.0]))
    assert sol["status"] == "primal infeasible" and sol["iterations"] == meta["primal_infeasible"]["iterations"]
    assert np.allclose(sol["z"], meta["primal_infeasible"]["z"], atol=1e-7) and sol["x"] is None
    G = spmatrix([-1.0, -1.0], [0, 1], [0, 1], (2, 2))
    sol = lp.conelp(np.array([-1.0, 0.5]), G, np.array([0.0, 0.0]))
    assert sol["status"] == "dual infeasible" and sol["iterations"] == meta["dual_infeasible"]["iterations"]
    assert np.allclose(sol["x"], meta["dual_infeasible"]["x"], rtol=1e-6) and sol["z"] is None
    with pytest.raises(ValueError):                                     # ml < n: Rank([G; A]) < n (coneprog.py:572-573)
        lp.conelp(np.zeros(3), spmatrix([1.0], [0], [0], (1, 3)), np.ones(1))


@pytest.mark.parametrize("name,gx,gy", [("qp6x5", 6, 5), ("qp25x20", 25, 20)])
def test_coneqp_golden(golden_dir, name, gx, gy):
    """G5: the reference's coneqp on the grid QP (S = P + G' W^-1 W^-T G refactored every iteration): same
    iteration count, same solution and objectives."""
    g = np.load(os.path.join(golden_dir, "g5_coneqp.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g5_coneqp.json")))["cases"][name]
    Q = workloads.qp_grid(gx, gy)
    G = spmatrix.from_ccs(Q["ml"], Q["n"], Q["Gp"], Q["Gi"], Q["Gx"])
    P = spmatrix.from_ccs(Q["n"], Q["n"], Q["Pp"], Q["Pi"], Q["Px"])
    sol = lp.coneqp(P, Q["q"], G, Q["h"])
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    assert rel(sol["x"], g[name + "_x"]) < 1e-7 and rel(sol["s"], g[name + "_s"]) < 1e-6 and rel(sol["z"], g[name + "_z"]) < 1e-6
    for key in ("primal objective", "dual objective"):
        assert abs(sol[key] - meta[key]) < 1e-8 * max(1.0, abs(meta[key]))
    assert sol["factorizations"] == meta["iterations"] + 1
    # iterative refinement changes nothing essential on a well-conditioned problem
    sol2 = lp.coneqp(P, Q["q"], G, Q["h"], {"refinement": 1})
    assert sol2["status"] == "optimal" and rel(sol2["x"], g[name + "_x"]) < 1e-7
    # KKT conditions of the result, checked on the host: P x + q + G' z = 0, G x + s = h, s, z >= 0, s'z small
    cols = np.repeat(np.arange(Q["n"]), np.diff(Q["Gp"]))
    Gx_ = np.zeros(Q["ml"]); np.add.at(Gx_, Q["Gi"], Q["Gx"] * sol["x"][cols])
    Gtz = np.zeros(Q["n"]); np.add.at(Gtz, cols, Q["Gx"] * sol["z"][Q["Gi"]])
    Px_ = workloads.sym_matvec(Q["n"], Q["Pp"], Q["Pi"], Q["Px"], sol["x"].reshape(-1, 1)).ravel()
    assert np.linalg.norm(Px_ + Q["q"] + Gtz) < 1e-6 * max(1.0, np.linalg.norm(Q["q"]))
    assert np.linalg.norm(Gx_ + sol["s"] - Q["h"]) < 1e-6 * max(1.0, np.linalg.norm(Q["h"]))
    assert sol["s"].min() > 0 and sol["z"].min() > 0 and sol["s"] @ sol["z"] < 1e-4


@pytest.mark.parametrize("name,gx,gy,p", [("qpeq6x5p4", 6, 5, 4), ("qpeq15x12p20", 15, 12, 20)])
def test_coneqp_with_equalities_golden(golden_dir, name, gx, gy, p):
    """G9: the reference's coneqp with equality constraints (misc.kkt_chol2 with H = P and K = A S^-1 A'): same
    iteration count, same x, y, s, z and objectives; with and without iterative refinement."""
    g = np.load(os.path.join(golden_dir, "g9_coneqp_eq.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g9_coneqp_eq.json")))["cases"][name]
    Q = workloads.qp_grid(gx, gy)
    L = workloads.lp_grid_eq(gx, gy, p)
    G = spmatrix.from_ccs(Q["ml"], Q["n"], Q["Gp"], Q["Gi"], Q["Gx"])
    P = spmatrix.from_ccs(Q["n"], Q["n"], Q["Pp"], Q["Pi"], Q["Px"])
    A = spmatrix.from_ccs(p, Q["n"], L["Ap"], L["Ai"], L["Ax"])
    sol = lp.coneqp(P, Q["q"], G, Q["h"], A=A, b=L["b"])
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    for k, tol in (("x", 1e-6), ("y", 1e-5), ("s", 1e-5), ("z", 1e-5)):
        assert rel(sol[k], g[name + "_" + k]) < tol, k
    for key in ("primal objective", "dual objective"):
        assert abs(sol[key] - meta[key]) < 1e-7 * max(1.0, abs(meta[key]))
    sol2 = lp.coneqp(P, Q["q"], G, Q["h"], {"refinement": 1}, A=A, b=L["b"])
    assert sol2["status"] == "optimal" and rel(sol2["x"], g[name + "_x"]) < 1e-6
    acols = np.repeat(np.arange(Q["n"]), np.diff(L["Ap"]))
    Ax_ = np.zeros(p); np.add.at(Ax_, L["Ai"], L["Ax"] * sol["x"][acols])
    assert np.linalg.norm(Ax_ - L["b"]) < 1e-7 * max(1.0, np.linalg.norm(L["b"]))


@pytest.mark.parametrize("name", ["all", "xs"])
def test_coneqp_with_initial_values_golden(golden_dir, name):
    """G11: coneqp(initvals=...) (coneprog.py:2108-2150): same iterations and solution as the reference."""
    g = np.load(os.path.join(golden_dir, "g11_coneqp_initvals.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g11_coneqp_initvals.json")))["cases"][name]
    Q = workloads.qp_grid(15, 12)
    L = workloads.lp_grid(15, 12)
    G = spmatrix.from_ccs(Q["ml"], Q["n"], Q["Gp"], Q["Gi"], Q["Gx"])
    P = spmatrix.from_ccs(Q["n"], Q["n"], Q["Pp"], Q["Pi"], Q["Px"])
    iv = {"x": L["x0"], "s": L["s0"]}
    if name == "all":
        iv["z"] = L["z0"]
    sol = lp.coneqp(P, Q["q"], G, Q["h"], initvals=iv)
    assert sol["status"] == meta["status"] == "optimal" and sol["iterations"] == meta["iterations"]
    for k in "xsz":
        assert rel(sol[k], g[name + "_" + k]) < 1e-6, k
    with pytest.raises(ValueError):
        lp.coneqp(P, Q["q"], G, Q["h"], initvals={"s": -L["s0"]})


def test_solvers_module_names(golden_dir, capsys):
    """kvxopt.solvers spelling: lp / qp / conelp / coneqp and the module-level options dict; progress output is on by default
    as in the reference (coneprog.py:456): the iteration table (:917-923) and the closing line (:941, :961)."""
    from kvxopt_amd import solvers
    meta = json.load(open(os.path.join(golden_dir, "g4_conelp.json")))["cases"]["grid6x5"]
    P = workloads.lp_grid(6, 5)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    capsys.readouterr()
    sol = solvers.lp(P["c"], G, P["h"])
    out = capsys.readouterr().out.splitlines()
    assert sol["status"] == "optimal" and sol["iterations"] == meta["iterations"]
    assert out[0].split() == ["pcost", "dcost", "gap", "pres", "dres", "k/t"] and out[1].startswith(" 0: ")
    assert len(out) == sol["iterations"] + 3 and out[-1] == "Optimal solution found."
    solvers.options["maxiters"] = 3
    try:
        assert solvers.conelp(P["c"], G, P["h"])["status"] == "unknown"          # maxiters reached (coneprog.py:940-960)
        assert capsys.readouterr().out.splitlines()[-1] == "Terminated (maximum number of iterations reached)."
        assert solvers.lp(P["c"], G, P["h"], options={"maxiters": 50, "show_progress": False})["status"] == "optimal"
        assert capsys.readouterr().out == ""
    finally:
        solvers.options.clear()
    Q = workloads.qp_grid(6, 5)
    Gq = spmatrix.from_ccs(Q["ml"], Q["n"], Q["Gp"], Q["Gi"], Q["Gx"])
    Pq = spmatrix.from_ccs(Q["n"], Q["n"], Q["Pp"], Q["Pi"], Q["Px"])
    mq = json.load(open(os.path.join(golden_dir, "g5_coneqp.json")))["cases"]["qp6x5"]
    sq = solvers.qp(Pq, Q["q"], Gq, Q["h"])
    assert sq["status"] == "optimal" and sq["iterations"] == mq["iterations"]
    with pytest.raises(NotImplementedError):
        solvers.lp(P["c"], G, P["h"], kktsolver="ldl")


def test_coneqp_errors():
    Q = workloads.qp_grid(6, 5)
    G = spmatrix.from_ccs(Q["ml"], Q["n"], Q["Gp"], Q["Gi"], Q["Gx"])
    P = spmatrix.from_ccs(Q["n"], Q["n"], Q["Pp"], Q["Pi"], Q["Px"])
    with pytest.raises(TypeError):
        lp.coneqp(P, Q["q"][:-1], G, Q["h"])
    with pytest.raises(TypeError):
        lp.coneqp(spmatrix([1.0], [0], [0], (2, 2)), Q["q"], G, Q["h"])


@pytest.mark.parametrize("name,gx,gy", [("std6x5", 6, 5), ("std15x12", 15, 12)])
def test_conelp_standard_form_golden(golden_dir, name, gx, gy):
    """G6: the reference's conelp on the standard-form grid LP of config 4a (A x = b, x >= 0): the equality branch
    of misc.kkt_chol2 (K = A S^-1 A', misc.py:1483-1487, 1545) -- same iteration count, same x, y, s, z."""
    g = np.load(os.path.join(golden_dir, "g6_conelp_std.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g6_conelp_std.json")))["cases"][name]
    L = workloads.lp_grid_std(gx, gy)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    for k, tol in (("x", 1e-6), ("y", 1e-6), ("s", 1e-6), ("z", 1e-6)):
        assert rel(sol[k], g[name + "_" + k]) < tol, k
    assert abs(sol["primal objective"] - meta["primal objective"]) < 1e-8 * abs(meta["primal objective"])
    assert sol["factorizations"] == meta["iterations"] + 1
    # feasibility on the host
    acols = np.repeat(np.arange(L["n"]), np.diff(L["Ap"]))
    Ax_ = np.zeros(L["p"]); np.add.at(Ax_, L["Ai"], L["Ax"] * sol["x"][acols])
    assert np.linalg.norm(Ax_ - L["b"]) < 1e-6 * np.linalg.norm(L["b"]) and sol["x"].min() > -1e-7


@pytest.mark.parametrize("name,gx,gy,p", [("eq6x5p4", 6, 5, 4), ("eq15x12p20", 15, 12, 20)])
def test_conelp_general_g_with_equalities_golden(golden_dir, name, gx, gy, p):
    """G8: the reference's conelp with a general sparse G and equality constraints: the branch of misc.kkt_chol2 with
    K = A S^-1 A' for a non-diagonal S (misc.py:1476-1487) -- on the device K is a dense p x p front (lp.KKTGenEqDev)."""
    g = np.load(os.path.join(golden_dir, "g8_conelp_eq.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g8_conelp_eq.json")))["cases"][name]
    L = workloads.lp_grid_eq(gx, gy, p)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    for k, tol in (("x", 1e-6), ("y", 1e-6), ("s", 1e-6), ("z", 1e-6)):
        assert rel(sol[k], g[name + "_" + k]) < tol, k
    assert abs(sol["primal objective"] - meta["primal objective"]) < 1e-8 * abs(meta["primal objective"])
    acols = np.repeat(np.arange(L["n"]), np.diff(L["Ap"]))
    Ax_ = np.zeros(L["p"]); np.add.at(Ax_, L["Ai"], L["Ax"] * sol["x"][acols])
    assert np.linalg.norm(Ax_ - L["b"]) < 1e-6 * max(1.0, np.linalg.norm(L["b"]))


def test_conelp_general_g_with_equalities_at_scale():
    """Config-4b grid (ml = 200 000, n = 50 000) with 200 equality rows: optimal, feasible, duality gap closed."""
    L = workloads.lp_grid_eq(250, 200, 200)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    assert sol["status"] == "optimal"
    assert sol["relative gap"] < 1e-6 and sol["primal infeasibility"] < 1e-7 and sol["dual infeasibility"] < 1e-7
    acols = np.repeat(np.arange(L["n"]), np.diff(L["Ap"]))
    Ax_ = np.zeros(L["p"]); np.add.at(Ax_, L["Ai"], L["Ax"] * sol["x"][acols])
    assert np.linalg.norm(Ax_ - L["b"]) < 1e-6 * max(1.0, np.linalg.norm(L["b"]))


@pytest.mark.parametrize("name", ["primal", "dual", "both"])
def test_conelp_with_user_starting_points_golden(golden_dir, name):
    """G10: primalstart / dualstart (coneprog.py:683-737, 806-842): same iteration count and solution as the reference;
    a start outside the cone is refused like there."""
    g = np.load(os.path.join(golden_dir, "g10_conelp_starts.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g10_conelp_starts.json")))["cases"][name]
    P = workloads.lp_grid(15, 12)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    kw = {}
    if name in ("primal", "both"):
        kw["primalstart"] = {"x": P["x0"], "s": P["s0"]}
    if name in ("dual", "both"):
        kw["dualstart"] = {"z": P["z0"]}
    sol = lp.conelp(P["c"], G, P["h"], **kw)
    assert sol["status"] == meta["status"] == "optimal" and sol["iterations"] == meta["iterations"]
    for k in "xsz":
        assert rel(sol[k], g[name + "_" + k]) < 1e-6, k
    assert abs(sol["primal objective"] - meta["primal objective"]) < 1e-8 * abs(meta["primal objective"])
    with pytest.raises(ValueError):
        lp.conelp(P["c"], G, P["h"], primalstart={"x": P["x0"], "s": -P["s0"]})
    with pytest.raises(ValueError):
        lp.conelp(P["c"], G, P["h"], dualstart={"z": np.zeros(P["ml"])})


def test_interior_point_runs_are_bitwise_reproducible():
    """No atomics anywhere on the path (mat-vecs are row gathers, reductions fixed trees, extend-add parent-pull):
    two runs of the same problem give identical bits."""
    L = workloads.lp_grid_eq(15, 12, 20)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    a = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    b = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    for k in "xysz":
        assert np.array_equal(a[k], b[k]), k
    P = workloads.lp_grid(25, 20)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    a = lp.conelp(P["c"], G, P["h"]); b = lp.conelp(P["c"], G, P["h"])
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["z"], b["z"]) and a["gap"] == b["gap"]


def test_conelp_equality_errors():
    L = workloads.lp_grid_std(6, 5)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    with pytest.raises(TypeError):
        lp.conelp(L["c"], G, L["h"], A=A, b=L["b"][:-1])
    with pytest.raises(NotImplementedError):
        lp.conelp(L["c"], G, L["h"], dims={"l": 0, "q": [L["ml"]], "s": []})


def test_cholmod_module_api():
    """Error behaviour and option handling of the kvxopt.cholmod mirror (cholmod.c error macros)."""
    A = spmatrix([10, 3, 5, -2, 5, 2], [0, 2, 1, 3, 2, 3], [0, 0, 1, 1, 2, 3])     # spsolvers.rst:556
    X = matrix(np.arange(8.0), (4, 2))
    cholmod.linsolve(A, X)
    assert np.allclose(X.a, [[-0.14634146, 0.04878049], [1.33333333, 4.0], [0.48780488, 1.17073171], [2.83333333, 7.5]])
    F = cholmod.symbolic(A)
    with pytest.raises(ValueError):
        cholmod.solve(F, matrix(np.ones(4)))                   # symbolic factor (cholmod.c:452-453)
    cholmod.numeric(A, F)
    assert abs(2.0 * np.sum(np.log(cholmod.diag(F)._a)) - 5.50533153593) < 1e-10        # spsolvers.rst:765
    Xi = cholmod.splinsolve(A, spmatrix(1.0, range(4), range(4)))
    assert np.allclose(Xi.todense()[0, 0], 1.22e-01, atol=5e-4)
    L = cholmod.getfactor(F)
    Ld = L.todense()
    p = F.fac.perm()
    Afull = A.todense() + np.tril(A.todense(), -1).T
    assert np.allclose(Ld @ Ld.T, Afull[np.ix_(p, p)], atol=1e-12)
    # options['supernodal'] = 0: P A P' = L D L'; the documented known answer of spsolvers.rst:766-772
    cholmod.options["supernodal"] = 0
    try:
        F0 = cholmod.symbolic(A)
        cholmod.numeric(A, F0)
        Di = matrix(1.0, (4, 1))
        cholmod.solve(F0, Di, sys=6)
        assert abs(-np.sum(np.log(Di._a)) - 5.50533153593) < 1e-10
        with pytest.raises(ValueError):
            cholmod.diag(F0)                                   # cholmod.c:919-922: not a supernodal LL' factor
        cholmod.options["supernodal"] = 3
        with pytest.raises(ValueError):
            cholmod.symbolic(A)
    finally:
        cholmod.options.clear()
    cholmod.options["bogus"] = 1
    with pytest.raises(ValueError):
        cholmod.symbolic(A)                                    # cholmod.c:118-124
    cholmod.options.clear()
    with pytest.raises(TypeError):
        cholmod.symbolic(spmatrix([1.0], [0], [0], (2, 3)))
    with pytest.raises(ValueError):
        cholmod.symbolic(A, uplo="X")
    with pytest.raises(ValueError):
        cholmod.symbolic(A, p=matrix([0, 0, 1, 2], tc="i"))
    bad = spmatrix([1.0, 2.0, 1.0], [0, 1, 1], [0, 0, 1])
    Fb = cholmod.symbolic(bad)
    with pytest.raises(ArithmeticError) as e:
        cholmod.numeric(bad, Fb)
    assert e.value.args[0] == 1
    with pytest.raises(ArithmeticError):
        cholmod.solve(Fb, matrix(np.ones(2)))
    with pytest.raises(ValueError):
        cholmod.diag(Fb)


def test_lp_full_size_properties():
    """BASELINE.json configs[3] (4b, ml = 200 000, n = 50 000): KKT-step properties without an oracle
    run -- assembled S is SPD and the KKT solve satisfies the reduced system to 1e-10."""
    P = workloads.lp_grid(250, 200)
    ml, n = P["ml"], P["n"]
    rng = np.random.default_rng(4)
    di = rng.uniform(0.2, 5.0, ml)
    kkt = lp.KKTChol2Dev(ml, n, P["Gp"], P["Gi"], P["Gx"])
    kkt.factor(lp.DVec(ml, di))
    bx, bz = rng.standard_normal(n), rng.standard_normal(ml)
    xv, zv = lp.DVec(n, bx), lp.DVec(ml, bz)
    kkt.solve(xv, zv)
    ux, wz = xv.get(), zv.get()
    cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
    Gx_ = np.zeros(ml); np.add.at(Gx_, P["Gi"], P["Gx"] * ux[cols])
    uz = wz * di                                                   # uz = W^{-1} (W uz)
    # second block row: G ux - W'W uz = bz
    assert np.linalg.norm(Gx_ - uz / (di * di) - bz) / np.linalg.norm(bz) < 1e-10
    Gtz = np.zeros(n); np.add.at(Gtz, cols, P["Gx"] * uz[P["Gi"]])
    assert np.linalg.norm(Gtz - bx) / np.linalg.norm(bx) < 1e-9   # first block row: G' uz = bx



@pytest.mark.parametrize("mode", [0, 1])
def test_cholmod_ldl_factor_semantics(mode):
    """options['supernodal'] = 0 (and 1 on a sparse factor: flops / nnz(L) < 40): P A P' = L D L' with unit lower L.
    getfactor returns L with D on its diagonal (what cholmod_factor_to_sparse gives for an LDL' factor); every `sys` code of
    solve / spsolve follows the table of spsolvers.rst:640-668 with that L and D -- checked against dense numpy solves."""
    n, cp, ri, v = workloads.laplacian_2d(9, 7)
    A = spmatrix.from_ccs(n, n, cp, ri, v + 0.01 * np.random.default_rng(1).random(v.size) * (ri == np.repeat(np.arange(n), np.diff(cp))))
    Afull = A.todense() + np.tril(A.todense(), -1).T
    cholmod.options["supernodal"] = mode
    try:
        F = cholmod.symbolic(A)
        assert F.fac.info()["is_ll"] == 0
        cholmod.numeric(A, F)
        p = F.fac.perm()
        rng = np.random.default_rng(2)
        B = rng.standard_normal((n, 3))
        got = {}
        for sys in range(9):
            X = matrix(B.copy(order="F"))
            cholmod.solve(F, X, sys=sys)
            got[sys] = X.a.copy()
        Bs = spmatrix(rng.standard_normal(5), [0, 3, 7, 11, 20], [0, 0, 1, 1, 1], (n, 2))
        gsp = {sys: cholmod.spsolve(F, Bs, sys=sys).todense() for sys in range(9)}
        M = cholmod.getfactor(F).todense()                     # (getfactor leaves F symbolic in the reference too)
    finally:
        cholmod.options.clear()
    D = np.diag(np.diag(M))
    L = np.tril(M, -1) + np.eye(n)
    assert np.all(np.diag(M) > 0) and np.allclose(L @ D @ L.T, Afull[np.ix_(p, p)], atol=1e-12)
    P = np.eye(n)[p]                                           # (P x)_i = x_{p_i}
    ops = {0: Afull, 1: L @ D @ L.T, 2: L @ D, 3: D @ L.T, 4: L, 5: L.T, 6: D, 7: P.T, 8: P}
    for sys, Mop in ops.items():
        assert rel(got[sys], np.linalg.solve(Mop, B)) < 1e-11, sys
        assert rel(gsp[sys], np.linalg.solve(Mop, Bs.todense())) < 1e-11, sys


def test_base_gemv_sparse_and_dense_with_offsets():
    """base.gemv (base.c:744 -> sparse.c:1073-1104): y := alpha op(A) x + beta y with offsetx / offsety, sparse A and a dense
    A (taken as a matrix with every entry stored -- the same device kernel, no host arithmetic)."""
    rng = np.random.default_rng(21)
    m, n = 13, 7
    Ad = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.4)
    I, J = np.nonzero(Ad)
    As = spmatrix(Ad[I, J], I, J, (m, n))
    for A in (As, matrix(Ad)):
        for trans, lx, ly in (("N", n, m), ("T", m, n)):
            x = rng.standard_normal(lx + 3); y = rng.standard_normal(ly + 2)
            yx = matrix(y.copy())
            base.gemv(A, matrix(x), yx, trans=trans, alpha=-1.5, beta=0.5, offsetx=3, offsety=2)
            op = Ad if trans == "N" else Ad.T
            ref = y.copy(); ref[2:] = -1.5 * (op @ x[3:]) + 0.5 * y[2:]
            assert rel(yx._a, ref) < 1e-14



def test_base_gemv_subblocks_match_reference(golden_dir):
    """G13: base.gemv with m, n, offsetA, incx, incy (sparse.c:1073-1104: the m x n block of A at
    (offsetA % nrows, offsetA // nrows)), expected vectors from the reference's own base.gemv."""
    g = np.load(os.path.join(golden_dir, "g13_gemv_subblocks.npz"))
    A = spmatrix.from_ccs(int(g["M"]), int(g["N"]), g["cp"], g["ri"], g["v"])
    for ci, (trans, m, n, oA, incx, incy, ox, oy, alpha, beta) in enumerate(json.loads(str(g["cases"]))):
        y = matrix(g["c%d_y" % ci].copy())
        base.gemv(A, matrix(g["c%d_x" % ci]), y, trans=trans, alpha=alpha, beta=beta, m=m, n=n, incx=incx, incy=incy,
                  offsetA=oA, offsetx=ox, offsety=oy)
        assert np.allclose(y._a, g["c%d_out" % ci], rtol=1e-14, atol=1e-15), ci
    y = matrix(np.array([np.nan, 1.0, np.nan]))
    base.gemv(A, matrix(np.ones(2)), y, m=3, n=2, beta=0.0)       # beta = 0 overwrites y, whatever it held
    assert np.all(np.isfinite(y._a))
    with pytest.raises(TypeError):
        base.gemv(A, matrix(np.ones(7)), matrix(np.ones(9)), m=9, n=7, offsetA=1)


class _ForeignDense(array.array):
    """Stand-in for the reference's own `matrix`: nothing but .size, .typecode and the buffer protocol
    (dense.c:1350-1385 exports a column-major 'd' buffer)."""

    def __new__(cls, a):
        a = np.asarray(a, dtype=np.float64)
        self = super().__new__(cls, "d", a.reshape(-1, order="F").tolist())
        self.size = a.shape if a.ndim == 2 else (a.size, 1)
        return self


class _ForeignSparse:
    """Stand-in for the reference's own `spmatrix`: .size, .typecode and .CCS only (sparse.c: CCS getter returns three
    kvxopt matrices; plain sequences here)."""

    def __init__(self, n, cp, ri, v):
        self.size = (n, n)
        self.typecode = "d"
        self.CCS = (list(map(int, cp)), list(map(int, ri)), list(map(float, v)))


def test_cholmod_accepts_foreign_matrix_objects():
    """The drop-in claim of SURVEY 8(b): kvxopt_amd.cholmod driven with objects that are NOT this package's types -- only
    the attributes the reference's own matrix / spmatrix expose (.CCS, .size, .typecode, buffer protocol)."""
    n, cp, ri, v = workloads.laplacian_2d(13, 9)
    A = _ForeignSparse(n, cp, ri, v)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 2))
    F = cholmod.symbolic(A)
    cholmod.numeric(A, F)
    Xf = _ForeignDense(B)
    cholmod.solve(F, Xf)                            # overwritten in place through the buffer protocol (cholmod.c:482-491)
    X = np.array(Xf).reshape(n, 2, order="F")
    R = workloads.sym_matvec(n, cp, ri, v, X) - B
    assert np.linalg.norm(R) / np.linalg.norm(B) < 1e-12
    X2 = _ForeignDense(B)
    cholmod.linsolve(A, X2)
    assert rel(np.array(X2).reshape(n, 2, order="F"), X) < 1e-13
    d = cholmod.diag(F)
    assert np.all(np.asarray(d._a) > 0)
    # misc.kkt_chol2 with a foreign sparse G: factor + solve equals a dense KKT solve
    ml, nn = 12, 5
    Gd = rng.standard_normal((ml, nn)) * (rng.random((ml, nn)) < 0.6) + np.vstack([np.eye(nn), np.zeros((ml - nn, nn))])
    I, J = np.nonzero(Gd)
    Gs = spmatrix(Gd[I, J], I, J, (ml, nn))

    class FG:
        size, typecode, CCS = Gs.size, "d", (Gs.colptr.tolist(), Gs.rowind.tolist(), Gs.values.tolist())
    dims = {"l": ml, "q": [], "s": []}
    factor = misc.kkt_chol2(FG(), dims, spmatrix([], [], [], (0, nn)))
    dd = rng.uniform(0.5, 2.0, ml)
    f = factor(W_of(dd, 1.0 / dd))
    bx, bz = rng.standard_normal(nn), rng.standard_normal(ml)
    x, z = matrix(bx.copy()), matrix(bz.copy())
    f(x, matrix(np.zeros((0, 1))), z)
    K = np.block([[np.zeros((nn, nn)), Gd.T], [Gd, -np.diag(dd * dd)]])
    sol = np.linalg.solve(K, np.concatenate([bx, bz]))
    assert rel(x._a, sol[:nn]) < 1e-10 and rel(z._a, dd * sol[nn:]) < 1e-10


def test_numeric_checks_the_analysed_triangle_only():
    """cholmod.c:137-157 (`pack` reads only the `uplo` triangle): a matrix with the same number of entries in OTHER positions
    is refused; one that differs only in the ignored triangle factors fine."""
    n, cp, ri, v = workloads.laplacian_2d(6, 5)
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    F = cholmod.symbolic(A)
    cholmod.numeric(A, F)
    ri2 = ri.copy()
    j = 3                                           # move one sub-diagonal entry of column 3 to another row
    p = cp[j] + 1
    ri2[p] = ri[p] + 1 if (p + 1 == cp[j + 1] or ri[p] + 1 < ri[p + 1]) else ri[p]
    assert not np.array_equal(ri2, ri)
    with pytest.raises(ValueError):
        cholmod.numeric(spmatrix.from_ccs(n, n, cp, ri2, v), F)
    # full symmetric storage with junk in the upper triangle: only the lower one is read
    Ad = A.todense()
    full = Ad + np.tril(Ad, -1).T * 7.0
    I, J = np.nonzero(full)
    Afull = spmatrix(full[I, J], I, J, (n, n))
    cholmod.numeric(Afull, F)
    b = np.random.default_rng(1).standard_normal(n)
    x = matrix(b.copy())
    cholmod.solve(F, x)
    S = Ad + np.tril(Ad, -1).T
    assert rel(S @ x._a, b) < 1e-12
    F2 = cholmod.symbolic(Afull)                    # analysed from the full matrix, refactored from the triangle
    cholmod.numeric(A, F2)
    x2 = matrix(b.copy()); cholmod.solve(F2, x2)
    assert rel(x2._a, x._a) < 1e-13


def test_kkt_chol2_singular_fallback_and_diagonal_S_golden(golden_dir):
    """G14 (pure reference, dense-G LAPACK branch): (i) S = G' W^-2 G singular because a variable appears in no inequality,
    the equality rows restore full rank -- the reference's fallback S + A'A (misc.py:1433-1447, 1525-1526), first call and a
    refactorisation; (ii) a G with one entry per row (diagonal S, the standard-form branch with K on a fixed pattern)."""
    g = np.load(os.path.join(golden_dir, "g14_kkt_singular.npz"))
    ml, n, p = 14, 6, 2
    G = spmatrix.from_ccs(ml, n, g["sing_G_cp"], g["sing_G_ri"], g["sing_G_v"])
    A = spmatrix.from_ccs(p, n, g["sing_A_cp"], g["sing_A_ri"], g["sing_A_v"])
    f = misc.kkt_chol2(G, {"l": ml, "q": [], "s": []}, A)
    for i, d in enumerate((np.ones(ml), g["sing_d"])):
        solve = f(W_of(d.copy(), 1.0 / d))
        x, y, z = matrix(g["sing_bx"].copy()), matrix(g["sing_by"].copy()), matrix(g["sing_bz"].copy())
        solve(x, y, z)
        assert rel(x._a, g["sing_x%d" % i]) < 1e-10 and rel(y._a, g["sing_y%d" % i]) < 1e-10 and rel(z._a, g["sing_z%d" % i]) < 1e-10, i
    n2, p2 = 7, 3
    ml2 = 2 * n2
    G2 = spmatrix.from_ccs(ml2, n2, g["diag_G_cp"], g["diag_G_ri"], g["diag_G_v"])
    A2 = spmatrix.from_ccs(p2, n2, g["diag_A_cp"], g["diag_A_ri"], g["diag_A_v"])
    f2 = misc.kkt_chol2(G2, {"l": ml2, "q": [], "s": []}, A2)
    f2(W_of(np.ones(ml2), np.ones(ml2)))
    solve = f2(W_of(g["diag_d"].copy(), 1.0 / g["diag_d"]))
    x, y, z = matrix(g["diag_bx"].copy()), matrix(g["diag_by"].copy()), matrix(g["diag_bz"].copy())
    solve(x, y, z)
    assert rel(x._a, g["diag_x"]) < 1e-10 and rel(y._a, g["diag_y"]) < 1e-10 and rel(z._a, g["diag_z"]) < 1e-10
    with pytest.raises(ArithmeticError):                   # p = 0: a singular S has no fallback (misc.py:1433 needs A)
        misc.kkt_chol2(G, {"l": ml, "q": [], "s": []}, spmatrix([], [], [], (0, n)))(W_of(np.ones(ml), np.ones(ml)))


def test_kkt_general_S_with_many_equality_rows():
    """K = A S^-1 A' for a non-diagonal S and MORE equality rows than one column block of X = S^-1 A' holds (the p <= 2048
    limit of round 1 is gone: X is formed and consumed block by block, K is a dense p x p front).  Checked against a sparse
    direct solve of the whole KKT system on the host (SciPy), residual and solution."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    P = workloads.lp_grid(40, 30)
    ml, n = P["ml"], P["n"]
    p = 300
    rng = np.random.default_rng(12)
    Ad = sp.random(p, n, density=3.0 / n, random_state=5, format="csc") + sp.csc_matrix((np.ones(p), (np.arange(p), rng.permutation(n)[:p])), shape=(p, n))
    Ad = Ad.tocsc(); Ad.sort_indices()
    old = lp.KKTGenEqDev.BLOCK_BYTES
    lp.KKTGenEqDev.BLOCK_BYTES = 8 * n * 64               # 64 columns per block: five blocks for p = 300
    try:
        kkt = lp.KKTGenEqDev(ml, n, P["Gp"], P["Gi"], P["Gx"], p, Ad.indptr.astype(np.int64), Ad.indices.astype(np.int64), Ad.data)
        assert kkt.cols == 64
        d = rng.uniform(0.5, 2.0, ml)
        kkt.factor(lp.DVec(ml, 1.0 / d))
        bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(ml)
        x, y, z = lp.DVec(n, bx), lp.DVec(p, by), lp.DVec(ml, bz)
        kkt.solve(x, y, z)
    finally:
        lp.KKTGenEqDev.BLOCK_BYTES = old
    G = sp.csc_matrix((P["Gx"], P["Gi"], P["Gp"]), shape=(ml, n))
    K = sp.bmat([[None, Ad.T, G.T], [Ad, None, None], [G, None, -sp.diags(d * d)]], format="csc")
    K = K + sp.csc_matrix((n + p + ml, n + p + ml))
    sol = spla.spsolve(K.tocsc(), np.concatenate([bx, by, bz]))
    ux, uy, uz = sol[:n], sol[n:n + p], sol[n + p:]
    assert rel(x.get(), ux) < 1e-8 and rel(y.get(), uy) < 1e-8 and rel(z.get(), d * uz) < 1e-8


@pytest.mark.parametrize("tag,mnl", [("a", None), ("b", 3)])
def test_second_order_cone_scaling_golden(golden_dir, tag, mnl):
    """G15 (pure reference): Nesterov-Todd scaling with second-order-cone blocks, dims = {'l': 4, 'q': [5, 1, 9]} without and
    with a nonlinear block -- compute_scaling (W['v'], W['beta'], lambda), scale in all four trans / inverse combinations on
    two columns, scale2, sprod, sinv, ssqr, sdot, max_step and update_scaling (s, z, lambda, v, beta in place).  Identical
    formulas; the inner products are summed in a different order than BLAS: 1e-13."""
    g = np.load(os.path.join(golden_dir, "g15_q_cone_scaling.npz"))
    k = mnl or 0
    ml, q = 4, [5, 1, 9]
    dims = {"l": ml, "q": q, "s": []}
    N = k + ml + sum(q)
    tol = 1e-13
    lm = matrix(0.0, (N, 1))
    W = misc.compute_scaling(matrix(g[tag + "_s"].copy()), matrix(g[tag + "_z"].copy()), lm, dims, mnl)
    assert rel(lm._a, g[tag + "_lmbda"]) < tol
    assert rel(W["d"]._a, g[tag + "_d"]) < tol and rel(W["di"]._a, g[tag + "_di"]) < tol
    if mnl:
        assert rel(W["dnl"]._a, g[tag + "_dnl"]) < tol and rel(W["dnli"]._a, g[tag + "_dnli"]) < tol
    assert len(W["v"]) == len(q) and [v.size for v in W["v"]] == [(m, 1) for m in q]
    assert rel(np.concatenate([v._a for v in W["v"]]), g[tag + "_v"]) < tol and rel(W["beta"], g[tag + "_beta"]) < tol
    for tr in "NT":
        for inv in "NI":
            x = matrix(g[tag + "_X"].copy(order="F"))
            misc.scale(x, W, trans=tr, inverse=inv)
            assert rel(x.a, g["%s_scale_%s%s" % (tag, tr, inv)]) < tol, (tr, inv)
    x1, y1 = g[tag + "_x1"], g[tag + "_y1"]
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, k); assert rel(a._a, g[tag + "_scale2_N"]) < tol
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, k, inverse="I"); assert rel(a._a, g[tag + "_scale2_I"]) < tol
    a = matrix(x1.copy()); misc.sprod(a, matrix(y1), dims, k); assert rel(a._a, g[tag + "_sprod"]) < tol
    a = matrix(x1.copy()); misc.sinv(a, matrix(y1), dims, k); assert rel(a._a, g[tag + "_sinv"]) < tol
    a = matrix(0.0, (N, 1)); misc.ssqr(a, matrix(y1), dims, k); assert rel(a._a, g[tag + "_ssqr"]) < tol
    assert abs(misc.sdot(matrix(x1), matrix(y1), dims, k) - float(g[tag + "_sdot"])) <= 1e-12 * abs(float(g[tag + "_sdot"]))
    assert abs(misc.max_step(matrix(x1), dims, k) - float(g[tag + "_max_step"])) <= 1e-13 * max(1.0, abs(float(g[tag + "_max_step"])))
    ms, mz = matrix(g[tag + "_us_s_in"].copy()), matrix(g[tag + "_us_z_in"].copy())
    misc.update_scaling(W, lm, ms, mz)
    for got, name in ((ms, "us_s"), (mz, "us_z"), (lm, "us_lmbda"), (W["d"], "us_d"), (W["di"], "us_di")):
        assert rel(got._a, g[tag + "_" + name]) < tol, name
    assert rel(np.concatenate([v._a for v in W["v"]]), g[tag + "_us_v"]) < tol and rel(W["beta"], g[tag + "_us_beta"]) < tol
    # inverse scaling undoes the scaling; W z = W^-T s = lambda holds for the 'q' blocks too
    x = matrix(g[tag + "_X"].copy(order="F"))
    misc.scale(x, W); misc.scale(x, W, inverse="I")
    assert rel(x.a, g[tag + "_X"]) < 1e-12
    assert misc.max_step(matrix(np.zeros(0)), {"l": 0, "q": [], "s": []}) == 0.0


def _col_signs(got, ref, m):
    """Singular / eigen-vectors are defined up to the sign of each column: flip the columns of `got` (m x m, flat column-major)
    to the orientation of `ref`."""
    G, R = got.reshape((m, m), order="F").copy(), ref.reshape((m, m), order="F")
    for j in range(m):
        if np.dot(G[:, j], R[:, j]) < 0:
            G[:, j] = -G[:, j]
    return G.reshape(-1, order="F")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mnl", [("a", None), ("b", 2)])
def test_s_cone_scaling_against_reference(golden_dir, tag, mnl):
    """Golden G16 (pure reference: misc.compute_scaling / update_scaling / ssqr, misc_solvers.scale / scale2 / sprod / sinv / sdot /
    max_step, pack / pack2 / unpack / symm / trisc / triusc) on dims = {'l': 3, 'q': [4], 's': [3, 1, 6]}, also with a nonlinear
    block.  The reference goes through LAPACK's gesvd / syevd, the GPU through one-sided Jacobi: singular values / eigenvalues
    agree to 1e-12, the vectors up to the sign of each column (compared after aligning the signs; the products r r', rti rti'
    -- the scaling itself -- are compared as they are).  Layout helpers are bit-exact."""
    g = np.load(os.path.join(golden_dir, "g16_s_cone_scaling.npz"))
    k = mnl or 0
    ml, q, sd = 3, [4], [3, 1, 6]
    dims = {"l": ml, "q": q, "s": sd}
    nlq = k + ml + sum(q)
    N = nlq + sum(m * m for m in sd)
    Nd = nlq + sum(sd)
    off2 = np.concatenate([[0], np.cumsum([m * m for m in sd])])
    tol = 1e-12
    lm = matrix(0.0, (Nd, 1))
    W = misc.compute_scaling(matrix(g[tag + "_s"].copy()), matrix(g[tag + "_z"].copy()), lm, dims, mnl)
    assert rel(lm._a, g[tag + "_lmbda"]) < tol
    assert [r.size for r in W["r"]] == [(m, m) for m in sd] and [r.size for r in W["rti"]] == [(m, m) for m in sd]
    for i, m in enumerate(sd):
        rr, rt = g[tag + "_r"][off2[i]:off2[i + 1]], g[tag + "_rti"][off2[i]:off2[i + 1]]
        R, Rr = W["r"][i].a, rr.reshape((m, m), order="F")
        T, Tr = W["rti"][i].a, rt.reshape((m, m), order="F")
        assert rel(R @ R.T, Rr @ Rr.T) < tol and rel(T @ T.T, Tr @ Tr.T) < tol
        assert rel(_col_signs(W["r"][i]._a, rr, m), rr) < 1e-10 and rel(_col_signs(W["rti"][i]._a, rt, m), rt) < 1e-10
        assert rel(R.T @ T, np.eye(m)) < tol                                   # rti = r^-T
        # the defining identities (misc.py:354-366): r' z r = r^-1 s r^-T = diag(lambda)
        ind = nlq + off2[i]
        Z = g[tag + "_z"][ind:ind + m * m].reshape((m, m), order="F")
        Sm = g[tag + "_s"][ind:ind + m * m].reshape((m, m), order="F")
        il = nlq + sum(sd[:i])
        assert rel(R.T @ Z @ R, np.diag(lm._a[il:il + m])) < 1e-11 and rel(T.T @ Sm @ T, np.diag(lm._a[il:il + m])) < 1e-11
    # x_k := r' X r lives in the basis r defines, so the goldens of scale are taken with the reference's own r, rti
    Wref = dict(W)
    Wref["r"] = [matrix(g[tag + "_r"][off2[i]:off2[i + 1]].copy(), (m, m)) for i, m in enumerate(sd)]
    Wref["rti"] = [matrix(g[tag + "_rti"][off2[i]:off2[i + 1]].copy(), (m, m)) for i, m in enumerate(sd)]
    for tr in "NT":
        for inv in "NI":
            x = matrix(g[tag + "_X"].copy(order="F"))
            misc.scale(x, Wref, trans=tr, inverse=inv)
            assert rel(x.a, g["%s_scale_%s%s" % (tag, tr, inv)]) < tol, (tr, inv)       # incl. the untouched upper triangles
    x = matrix(g[tag + "_X"].copy(order="F"))                  # with the GPU's own r: scaling and inverse scaling cancel
    misc.scale(x, W); misc.scale(x, W, inverse="I")
    low = np.concatenate([np.arange(nlq)] + [nlq + off2[i] + np.array([r + c * m for c in range(m) for r in range(c, m)])
                                             for i, m in enumerate(sd)])
    assert rel(x.a[low], g[tag + "_X"][low]) < 1e-11
    x1, y1, yd = g[tag + "_x1"], g[tag + "_y1"], g[tag + "_yd"]
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, k); assert rel(a._a, g[tag + "_scale2_N"]) < tol
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, k, inverse="I"); assert rel(a._a, g[tag + "_scale2_I"]) < tol
    a, b = matrix(x1.copy()), matrix(y1.copy())
    misc.sprod(a, b, dims, k)
    assert rel(a._a, g[tag + "_sprod"]) < tol and np.array_equal(b._a[nlq:], g[tag + "_sprod_y_after"][nlq:])
    a = matrix(x1.copy()); misc.sprod(a, matrix(yd), dims, k, diag="D"); assert rel(a._a, g[tag + "_sprod_D"]) < tol
    a = matrix(x1.copy()); misc.sinv(a, matrix(yd), dims, k); assert rel(a._a, g[tag + "_sinv"]) < tol
    a = matrix(0.0, (Nd, 1)); misc.ssqr(a, matrix(yd), dims, k); assert rel(a._a, g[tag + "_ssqr"]) < tol
    assert abs(misc.sdot(matrix(x1), matrix(y1), dims, k) - float(g[tag + "_sdot"])) <= 1e-12 * abs(float(g[tag + "_sdot"]))
    assert abs(misc.max_step(matrix(x1), dims, k) - float(g[tag + "_max_step"])) <= 1e-12 * max(1.0, abs(float(g[tag + "_max_step"])))
    xs, sg = matrix(x1.copy()), matrix(0.0, (sum(sd), 1))
    t = misc.max_step(xs, dims, k, sg)
    assert abs(t - float(g[tag + "_max_step_sigma_t"])) <= 1e-12 * max(1.0, abs(t))
    assert np.max(np.abs(sg._a - g[tag + "_max_step_sigma"])) <= 1e-12 * max(1.0, np.max(np.abs(sg._a)))
    assert np.array_equal(xs._a[:nlq], x1[:nlq])
    for i, m in enumerate(sd):
        ind = nlq + off2[i]
        ref = g[tag + "_max_step_x"][ind:ind + m * m]
        assert rel(_col_signs(xs._a[ind:ind + m * m], ref, m), ref) < 1e-9
    # storage helpers: pure data movement and one scaling per entry -- bit for bit
    npk = nlq + sum(m * (m + 1) // 2 for m in sd)
    yp = matrix(0.0, (npk + 3, 1)); misc.pack(matrix(x1), yp, dims, k, 0, 2); assert np.array_equal(yp._a, g[tag + "_pack"])
    yu = matrix(7.0, (N + 1, 1)); misc.unpack(yp, yu, dims, k, 2, 1); assert np.array_equal(yu._a, g[tag + "_unpack"])
    x2 = matrix(np.column_stack([x1, y1]).copy(order="F")); misc.pack2(x2, dims, k); assert np.array_equal(x2.a, g[tag + "_pack2"])
    if mnl is None:
        a = matrix(x1.copy()); misc.trisc(a, dims); assert np.array_equal(a._a, g[tag + "_trisc"])
        a = matrix(x1.copy()); misc.triusc(a, dims); assert np.array_equal(a._a, g[tag + "_triusc"])
        a = matrix(x1.copy()); misc.symm(a, 6, nlq + 10); assert np.array_equal(a._a, g[tag + "_symm"])
    # update_scaling: lambda and the scaling (r r', rti rti') against the reference; s_k, z_k leave as U and V' (up to signs)
    ms, mz = matrix(g[tag + "_us_s_in"].copy()), matrix(g[tag + "_us_z_in"].copy())
    lm = matrix(g[tag + "_lmbda"].copy())
    misc.update_scaling(Wref, lm, ms, mz)
    W = Wref
    assert rel(lm._a, g[tag + "_us_lmbda"]) < tol
    assert rel(ms._a[:nlq], g[tag + "_us_s"][:nlq]) < 1e-13 and rel(mz._a[:nlq], g[tag + "_us_z"][:nlq]) < 1e-13
    for i, m in enumerate(sd):
        rr, rt = g[tag + "_us_r"][off2[i]:off2[i + 1]], g[tag + "_us_rti"][off2[i]:off2[i + 1]]
        R, Rr = W["r"][i].a, rr.reshape((m, m), order="F")
        T, Tr = W["rti"][i].a, rt.reshape((m, m), order="F")
        assert rel(R @ R.T, Rr @ Rr.T) < 1e-11 and rel(T @ T.T, Tr @ Tr.T) < 1e-11
        assert rel(R.T @ T, np.eye(m)) < 1e-11
        ind = nlq + off2[i]
        U, Ur = ms._a[ind:ind + m * m], g[tag + "_us_s"][ind:ind + m * m]
        assert rel(_col_signs(U, Ur, m), Ur) < 1e-9
        Vt = mz._a[ind:ind + m * m].reshape((m, m), order="F")
        Vtr = g[tag + "_us_z"][ind:ind + m * m].reshape((m, m), order="F")
        assert rel(np.abs(np.sum(Vt * Vtr, axis=1)), np.ones(m)) < 1e-9           # rows of V' match up to sign
    # a block that is not positive definite: lapack.potrf's ArithmeticError
    bad = g[tag + "_s"].copy(); bad[nlq] = -1.0
    with pytest.raises(ArithmeticError):
        misc.compute_scaling(matrix(bad), matrix(g[tag + "_z"].copy()), matrix(0.0, (Nd, 1)), dims, mnl)


@pytest.mark.gpu
def test_s_cone_scaling_larger_blocks_against_numpy():
    """'s' blocks of orders the Jacobi sweeps have to work for (odd, one wavefront, more than one wavefront of rows): lambda
    against numpy's SVD of Lz' Ls, the defining identities of the scaling, max_step against numpy's eigvalsh."""
    rng = np.random.default_rng(77)
    sd = [33, 64, 1, 90]
    dims = {"l": 0, "q": [], "s": sd}
    blocks_s, blocks_z = [], []
    for m in sd:
        for lst in (blocks_s, blocks_z):
            B = rng.standard_normal((m, m))
            lst.append(B @ B.T / m + np.diag(rng.uniform(0.2, 1.5, m)))
    s = np.concatenate([b.reshape(-1, order="F") for b in blocks_s])
    z = np.concatenate([b.reshape(-1, order="F") for b in blocks_z])
    lm = matrix(0.0, (sum(sd), 1))
    W = misc.compute_scaling(matrix(s), matrix(z), lm, dims)
    il = 0
    for i, m in enumerate(sd):
        Ls, Lz = np.linalg.cholesky(blocks_s[i]), np.linalg.cholesky(blocks_z[i])
        sv = np.linalg.svd(Lz.T @ Ls, compute_uv=False)
        assert rel(lm._a[il:il + m], sv) < 1e-12
        R, T = W["r"][i].a, W["rti"][i].a
        assert rel(R.T @ blocks_z[i] @ R, np.diag(sv)) < 1e-10 and rel(T.T @ blocks_s[i] @ T, np.diag(sv)) < 1e-10
        assert rel(R.T @ T, np.eye(m)) < 1e-10
        il += m
    off2 = np.concatenate([[0], np.cumsum([m * m for m in sd])])
    # W z = W^-T s = lambda (coneprog.py:1031-1043 relies on it): scale(z, W) and scale(s, W, trans='T', inverse='I') give
    # diag(lambda_k) in every block (lower triangles)
    zz, ss = matrix(z.copy()), matrix(s.copy())
    misc.scale(zz, W)
    misc.scale(ss, W, trans="T", inverse="I")
    il = 0
    for i, m in enumerate(sd):
        for got in (zz, ss):
            Bk = np.tril(got._a[off2[i]:off2[i + 1]].reshape((m, m), order="F"))
            assert rel(Bk, np.diag(lm._a[il:il + m])) < 1e-10
        il += m
    x = np.concatenate([(0.5 * (B + B.T)).reshape(-1, order="F") for B in (rng.standard_normal((m, m)) for m in sd)])
    ev = [np.linalg.eigvalsh(x[off2[i]:off2[i + 1]].reshape((m, m), order="F")) for i, m in enumerate(sd)]
    t = misc.max_step(matrix(x), dims)
    assert abs(t - max(-e[0] for e in ev)) <= 1e-12 * max(1.0, abs(t))
    xs, sg = matrix(x.copy()), matrix(0.0, (sum(sd), 1))
    misc.max_step(xs, dims, 0, sg)
    assert np.max(np.abs(sg._a - np.concatenate(ev))) <= 1e-12 * np.max(np.abs(np.concatenate(ev)))
    for i, m in enumerate(sd):                                   # Q diag(sigma) Q' = x_k
        Q = xs._a[off2[i]:off2[i + 1]].reshape((m, m), order="F")
        X = x[off2[i]:off2[i + 1]].reshape((m, m), order="F")
        assert rel(Q @ np.diag(ev[i]) @ Q.T, X) < 1e-10 and rel(Q.T @ Q, np.eye(m)) < 1e-11


@pytest.mark.gpu
def test_s_cone_scaling_with_repeated_singular_values():
    """Degenerate spectra: s = z = I (every singular value of Lz' Ls equals 1: any orthonormal basis serves) and s = 4 z
    (W = 2 I): the scaling point r r' and lambda are still what the definition gives, the Jacobi sweeps terminate at once."""
    m = 7
    dims = {"l": 0, "q": [], "s": [m]}
    I = np.eye(m).reshape(-1, order="F")
    lm = matrix(0.0, (m, 1))
    W = misc.compute_scaling(matrix(I.copy()), matrix(I.copy()), lm, dims)
    assert rel(lm._a, np.ones(m)) < 1e-14 and rel(W["r"][0].a @ W["r"][0].a.T, np.eye(m)) < 1e-14
    rng = np.random.default_rng(3)
    B = rng.standard_normal((m, m))
    Z = B @ B.T + m * np.eye(m)
    W = misc.compute_scaling(matrix((4.0 * Z).reshape(-1, order="F")), matrix(Z.reshape(-1, order="F")), lm, dims)
    R = W["r"][0].a
    assert rel(R @ R.T, 2.0 * np.eye(m)) < 1e-12                    # W = r r' = (s z^-1)^(1/2) = 2 I
    assert rel(np.sort(lm._a), np.sort(2.0 * np.linalg.eigvalsh(Z))) < 1e-12
    t = misc.max_step(matrix(np.zeros(m * m)), dims)                 # the zero matrix: every eigenvalue 0
    assert t == 0.0


@pytest.mark.gpu
def test_conelp_reuses_kkt_objects_of_a_known_structure(monkeypatch):
    """lp.conelp / lp.coneqp keep the KKT device objects of the last constraint structures (lp._kkt_for): a second problem on
    the same patterns of G (and A, P) with OTHER values must be solved exactly as by a fresh object -- bit for bit, the runs
    being reproducible -- with or without equality rows, and clear_cache() must forget everything."""
    for build in (lambda: workloads.lp_grid(30, 24), lambda: workloads.lp_grid_eq(30, 24, 5)):
        P1 = build()
        # the second problem: the rows of G x <= h scaled by a positive diagonal (same patterns, other values, still feasible)
        P2 = dict(P1)
        dscale = np.random.default_rng(12).uniform(0.5, 2.0, P1["ml"])
        P2["Gx"] = P1["Gx"] * dscale[P1["Gi"]]
        P2["h"] = P1["h"] * dscale
        assert not np.array_equal(P1["Gx"], P2["Gx"])
        def run(P):
            G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
            if "p" in P and P["p"]:
                A = spmatrix.from_ccs(P["p"], P["n"], P["Ap"], P["Ai"], P["Ax"])
                return lp.conelp(P["c"], G, P["h"], A=A, b=P["b"])
            return lp.conelp(P["c"], G, P["h"])
        lp.clear_cache()
        run(P1)
        assert len(lp._KKT_CACHE) == 1
        warm = run(P2)                                   # cached structure, refreshed values
        assert len(lp._KKT_CACHE) == 1
        monkeypatch.setenv("KVX_LP_NO_CACHE", "1")
        cold = run(P2)
        monkeypatch.delenv("KVX_LP_NO_CACHE")
        assert warm["status"] == cold["status"] == "optimal" and warm["iterations"] == cold["iterations"]
        for key in ("x", "s", "z"):
            assert np.array_equal(np.asarray(warm[key]), np.asarray(cold[key])), key
    lp.clear_cache()
    assert len(lp._KKT_CACHE) == 0


@pytest.mark.gpu
def test_general_equality_kkt_solves_through_the_kept_solution_block(monkeypatch):
    """lp.KKTGenEqDev: with X = S^-1 A' of the last factorisation at hand (p columns fit one block), S^-1 (b - A' uy) is taken as
    S^-1 b - X uy -- one sweep through the factor of S per KKT system instead of two -- and the two systems of an iteration share
    their sweeps (two-column solves).  Same iterates as the two-sweep form (KVX_KKT_NO_X=1) to rounding: same iteration count,
    solution to 1e-8; the KKT residual of one solve is checked against the definition."""
    P = workloads.lp_grid_eq(40, 30, 24)
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
    A = spmatrix.from_ccs(P["p"], P["n"], P["Ap"], P["Ai"], P["Ax"])
    sols = {}
    for no_x in ("0", "1"):
        monkeypatch.setenv("KVX_KKT_NO_X", no_x)
        lp.clear_cache()
        sols[no_x] = lp.conelp(P["c"], G, P["h"], A=A, b=P["b"])
    monkeypatch.delenv("KVX_KKT_NO_X")
    lp.clear_cache()
    a, b = sols["0"], sols["1"]
    assert a["status"] == b["status"] == "optimal" and a["iterations"] == b["iterations"]
    for key in ("x", "y", "s", "z"):
        assert np.abs(np.asarray(a[key]) - np.asarray(b[key])).max() <= 1e-8 * max(1.0, np.abs(np.asarray(b[key])).max()), key
    # one KKT solve against the dense definition  [0 A' G'; A 0 0; G 0 -W'W] (ux, uy, uz') = (bx, by, bz),  W = diag(1 / di)
    ml, n, p = P["ml"], P["n"], P["p"]
    kkt = lp.KKTGenEqDev(ml, n, P["Gp"], P["Gi"], P["Gx"], p, P["Ap"], P["Ai"], P["Ax"])
    assert kkt.x_whole
    rng = np.random.default_rng(4)
    di = rng.uniform(0.5, 2.0, ml)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(ml)
    dv = lp.DVec(ml, di)
    kkt.factor(dv, sync=True)
    x, y, z = lp.DVec(n, bx), lp.DVec(p, by), lp.DVec(ml, bz)
    kkt.solve(x, y, z)
    ux, uy, wz = x.get(), y.get(), z.get()                      # z holds W uz
    uz = wz * di
    import scipy.sparse as sp
    Gs = sp.csc_matrix((P["Gx"], P["Gi"], P["Gp"]), shape=(ml, n)); As = sp.csc_matrix((P["Ax"], P["Ai"], P["Ap"]), shape=(p, n))
    r1 = As.T @ uy + Gs.T @ uz - bx
    r2 = As @ ux - by
    r3 = Gs @ ux - uz / di ** 2 - bz
    scale = max(np.abs(bx).max(), np.abs(by).max(), np.abs(bz).max())
    assert max(np.abs(r1).max(), np.abs(r2).max(), np.abs(r3).max()) < 1e-9 * scale * max(1.0, np.abs(uz).max())


@pytest.mark.gpu
def test_fused_iteration_is_bitwise():
    """Round 3 fused the ~85 short launches of an interior-point iteration into ~17 (kkt.hip "round 3": residuals, the vector
    operations around the KKT solves, the second half of f6_no_ir with its reductions, the update).  Same arithmetic, same
    roundings: every array and scalar a run returns is bit for bit that of the one-launch-per-operation path (KVX_LP_UNFUSED=1)."""
    import json
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipm_bits_probe.py")
    outs = []
    # ... and round 4 issues the fused launches of an iteration from C in four calls (kvx_lp_iter_*): the third arm is the round-3
    # form, one ctypes call per launch (KVX_LP_PYCALLS=1)
    for flags in ({"KVX_LP_UNFUSED": "0"}, {"KVX_LP_UNFUSED": "1"}, {"KVX_LP_UNFUSED": "0", "KVX_LP_PYCALLS": "1"}):
        env = dict(os.environ, **flags)
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and outs[0] == outs[2]
    assert all(v["status"] == "optimal" for v in outs[0].values())


@pytest.mark.gpu
def test_fused_iteration_entries_against_numpy():
    """The round-3 entries one by one against their definitions in numpy (misc.py:1489-1563, coneprog.py:861-896, 1250-1298,
    1339-1432; tolerance 1e-13 relative -- the bitwise comparison with the call-by-call path is test_fused_iteration_is_bitwise):
    kvx_kkt_solve_pre_dev / _post_dev with one and two right-hand sides and long rows, kvx_lp_residuals_dev, kvx_lp_update_x_dev,
    kvx_atda_assemble_sq_dev, kvx_lp_newton_rhs_dev without lmbdasq."""
    import ctypes
    import scipy.sparse as sp
    L = _lib.lib()
    rng = np.random.default_rng(7)
    ml, n = 700, 90
    G = sp.random(ml, n, density=0.06, random_state=3, format="csc") + sp.csc_matrix((rng.standard_normal(n), (rng.integers(0, ml, n), np.arange(n))), shape=(ml, n))
    G = sp.csc_matrix(G); G.sort_indices()
    assert np.diff(G.indptr).max() > 16 and np.diff(G.tocsr().indptr).max() > 8          # the 16-lane forms
    Gd = lp.SpMatDev(ml, n, G.indptr.astype(np.int64), G.indices.astype(np.int64), G.data)
    dv = lambda a: lp.DVec(len(a), a)
    close = lambda a, b: np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max())
    di = rng.uniform(0.5, 2.0, ml)
    xin = [rng.standard_normal(n) for _ in range(2)]
    zin = [rng.standard_normal(ml) for _ in range(2)]
    xs, xos, zos = [-1.0, 0.7], [0.3, 1.0], [0.3, 1.0]
    did = dv(di)
    for nrhs in (1, 2):
        xi, zi = [dv(a) for a in xin], [dv(a) for a in zin]
        xo, zo = [lp.DVec(n) for _ in range(2)], [lp.DVec(ml) for _ in range(2)]
        sides = lp._sides([(xi[k], xs[k], zi[k], xo[k], xos[k], zo[k], zos[k]) for k in range(nrhs)])
        x2 = lp.DVec(2 * n)
        _lib.raise_for(L.kvx_kkt_solve_pre_dev(ml, n, Gd.cp.ptr, Gd.ri.ptr, Gd.vx.ptr, Gd.max_col, did.ptr, nrhs, sides, x2.ptr, n))
        got = x2.get()
        for k in range(nrhs):
            assert close(got[k * n:(k + 1) * n], xs[k] * xin[k] + G.T @ (di * (zin[k] * di))), (nrhs, k)
        # (what the triangular solves would put there: anything -- the post step is a function of x2)
        x2v = rng.standard_normal(2 * n); x2.set(x2v)
        _lib.raise_for(L.kvx_kkt_solve_post_dev(ml, n, Gd.tcp.ptr, Gd.tri.ptr, Gd.tvx.ptr, Gd.max_row, did.ptr, nrhs, sides, x2.ptr, n))
        for k in range(nrhs):
            xk = x2v[k * n:(k + 1) * n]
            assert close(xo[k].get(), xos[k] * xk), (nrhs, k)
            assert close(zo[k].get(), zos[k] * (di * (G @ xk) - zin[k] * di)), (nrhs, k)
    # residuals
    x, z, s, c, h = rng.standard_normal(n), rng.uniform(0.1, 1, ml), rng.uniform(0.1, 1, ml), rng.standard_normal(n), rng.standard_normal(ml)
    tau = 0.8125
    out = [lp.DVec(n), lp.DVec(n), lp.DVec(ml), lp.DVec(ml)]
    ins = [dv(a) for a in (x, z, s, c, h)]                   # (kept alive: a freed vector's block goes back to the pool)
    _lib.raise_for(L.kvx_lp_residuals_dev(ml, n, Gd.cp.ptr, Gd.ri.ptr, Gd.vx.ptr, Gd.max_col, Gd.tcp.ptr, Gd.tri.ptr, Gd.tvx.ptr, Gd.max_row,
                                          *[a.ptr for a in ins], tau, *[o.ptr for o in out]))
    hrx = -(G.T @ z); hrz = G @ x + s
    for o, ref in zip(out, (hrx, hrx - tau * c, hrz, hrz - tau * h)):
        assert close(o.get(), ref)
    # update + x += step dx
    ds, dz, d, lm = rng.uniform(-0.5, 0.5, ml), rng.uniform(-0.5, 0.5, ml), rng.uniform(0.5, 2, ml), rng.uniform(0.5, 2, ml)
    dx, step = rng.standard_normal(n), 0.625
    v = [dv(a) for a in (ds, dz, d, np.zeros(ml), lm, np.zeros(ml), np.zeros(ml))]
    xd, dxd = dv(x), dv(dx)
    _lib.raise_for(L.kvx_lp_update_x_dev(ml, n, step, *[a.ptr for a in v], dxd.ptr, xd.ptr))
    ss, zz = np.sqrt((step * ds + 1) * lm), np.sqrt((step * dz + 1) * lm)
    dd = d * ss / zz
    for got, ref in zip(v, (ss, zz, dd, 1 / dd, ss * zz, ss * zz * dd, ss * zz / dd)):
        assert close(got.get(), ref)
    assert close(xd.get(), x + step * dx)
    # S = G' diag(di)^2 G with the square taken inside the assembly
    hpl = ctypes.c_void_p()
    _lib.raise_for(L.kvx_atda_plan(ml, n, _lib.pi(G.indptr.astype(np.int64)), _lib.pi(G.indices.astype(np.int64)), None, None, ctypes.byref(hpl)))
    snz = ctypes.c_int64()
    _lib.raise_for(L.kvx_atda_pattern(hpl, ctypes.byref(snz), None, None))
    Sp = np.empty(n + 1, dtype=np.int64); Si = np.empty(snz.value, dtype=np.int64)
    _lib.raise_for(L.kvx_atda_pattern(hpl, ctypes.byref(snz), _lib.pi(Sp), _lib.pi(Si)))
    Sx = lp.DVec(snz.value)
    _lib.raise_for(L.kvx_atda_assemble_sq_dev(hpl, Gd.vx.ptr, did.ptr, None, Sx.ptr))
    Sref = (G.T @ sp.diags(di * di) @ G).toarray()
    cols = np.repeat(np.arange(n), np.diff(Sp))
    assert close(Sx.get(), Sref[Si, cols])
    L.kvx_atda_free(hpl)
    # Newton right-hand side with lmbda o lmbda formed in the kernel
    rz, ws3 = rng.standard_normal(ml), rng.standard_normal(ml)
    o1, o2 = lp.DVec(ml), lp.DVec(ml)
    keep = [dv(a) for a in (ws3, rz, lm, d)]
    _lib.raise_for(L.kvx_lp_newton_rhs_dev(ml, None, keep[0].ptr, 0.25, 0.75, keep[1].ptr, keep[2].ptr, keep[3].ptr, o1.ptr, o2.ptr))
    dsr = -((lm * lm + ws3) - 0.25) / lm
    assert close(o1.get(), dsr) and close(o2.get(), -(0.75 * rz + dsr * d))


@pytest.mark.gpu
def test_config4a_full_size_standard_form():
    """BASELINE.json configs[3] in its literal form at full size (4a: 50 000 equality rows, 200 000 variables, x >= 0): the
    device-resident conelp through the equality branch (K = A S^-1 A' on a fixed pattern, lp.KKTDiagEqDev).  No oracle run at
    this size: optimality by its certificates -- primal and dual feasibility, complementarity, equal objectives -- checked on
    the host with the caller's own arrays (golden sizes with the reference's iterates: test_conelp_standard_form_golden)."""
    L = workloads.lp_grid_std(250, 200)
    n, p = L["n"], L["p"]
    assert (p, n) == (50000, 200000)
    G = spmatrix.from_ccs(L["ml"], n, L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(p, n, L["Ap"], L["Ai"], L["Ax"])
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    assert sol["status"] == "optimal" and sol["iterations"] <= 30
    x, y, s, z = sol["x"], sol["y"], sol["s"], sol["z"]
    acols = np.repeat(np.arange(n), np.diff(L["Ap"]))
    Ax_ = np.zeros(p); np.add.at(Ax_, L["Ai"], L["Ax"] * x[acols])
    assert np.linalg.norm(Ax_ - L["b"]) <= 1e-7 * np.linalg.norm(L["b"])             # A x = b
    assert x.min() > -1e-8 and np.abs(s - x).max() <= 1e-8 * max(1.0, np.abs(x).max())   # G = -I, h = 0: s = x >= 0
    Aty = L["Ax"] * y[L["Ai"]]                                                        # (A' y)_j = sum over column j
    Aty = np.add.reduceat(Aty, L["Ap"][:-1]) * (np.diff(L["Ap"]) > 0)
    assert z.min() > -1e-8 and np.linalg.norm(L["c"] + Aty - z) <= 1e-7 * np.linalg.norm(L["c"])   # c + A'y + G'z = 0, z >= 0
    assert abs(s @ z) <= 1e-6 * max(1.0, abs(L["c"] @ x))                              # complementarity
    assert abs(sol["primal objective"] - sol["dual objective"]) <= 1e-6 * max(1.0, abs(sol["primal objective"]))


@pytest.mark.gpu
@pytest.mark.parametrize("with_eq", [False, True])
def test_conelp_with_a_user_kktsolver(with_eq):
    """The reference's plug-in point kktsolver(W) -> f(x, y, z) (coneprog.py:323-344; its own test: tests/test_custom_kkt.py)
    under the device-resident conelp (lp.KKTUserHost): (1) a dense numpy solver of the 3 x 3 block system written against the
    documented contract -- on entry (bx, by, bz), on exit (ux, uy, W uz); (2) kvxopt_amd.misc.kkt_chol2, the host-array mirror of
    the reference's default.  Both reach the default path's solution in the same number of iterations."""
    from kvxopt_amd import solvers
    L = workloads.lp_grid_eq(9, 7, 5) if with_eq else workloads.lp_grid(9, 7)
    ml, n = L["ml"], L["n"]
    G = spmatrix.from_ccs(ml, n, L["Gp"], L["Gi"], L["Gx"])
    kw = {}
    Gd = np.zeros((ml, n)); Gd[L["Gi"], np.repeat(np.arange(n), np.diff(L["Gp"]))] = L["Gx"]
    p = 0
    Ad = np.zeros((0, n))
    if with_eq:
        p = L["p"]
        kw = {"A": spmatrix.from_ccs(p, n, L["Ap"], L["Ai"], L["Ax"]), "b": L["b"]}
        Ad = np.zeros((p, n)); Ad[L["Ai"], np.repeat(np.arange(n), np.diff(L["Ap"]))] = L["Ax"]
    ref = solvers.conelp(L["c"], G, L["h"], options={"show_progress": False}, **kw)
    calls = {"factor": 0, "solve": 0}

    def dense_kkt(W):
        d = np.asarray(W["d"]._a)
        assert W["di"].size == (ml, 1) and np.allclose(np.asarray(W["di"]._a) * d, 1.0)
        K = np.zeros((n + p + ml, n + p + ml))
        K[:n, n:n + p] = Ad.T; K[:n, n + p:] = Gd.T
        K[n:n + p, :n] = Ad; K[n + p:, :n] = Gd
        K[n + p:, n + p:] = -np.diag(d * d)
        lu = np.linalg.inv(K)
        calls["factor"] += 1

        def f(x, y, z):
            calls["solve"] += 1
            u = lu @ np.concatenate([x._a, y._a, z._a])
            x._a[:] = u[:n]
            y._a[:] = u[n:n + p]
            z._a[:] = d * u[n + p:]
        return f

    s1 = solvers.conelp(L["c"], G, L["h"], options={"show_progress": False}, kktsolver=dense_kkt, **kw)
    assert s1["status"] == "optimal" and s1["iterations"] == ref["iterations"]
    assert calls["factor"] == ref["iterations"] + 1 and calls["solve"] == 3 * ref["iterations"] + 2
    for k in ("x", "s", "z"):
        assert rel(s1[k], ref[k]) < 1e-6, k
    fac = misc.kkt_chol2(G, {"l": ml, "q": [], "s": []}, kw.get("A", spmatrix([], [], [], (0, n))))
    s2 = solvers.lp(L["c"], G, L["h"], options={"show_progress": False}, kktsolver=lambda W: fac(W), **kw)
    assert s2["status"] == "optimal" and s2["iterations"] == ref["iterations"]
    for k in ("x", "s", "z"):
        assert rel(s2[k], ref[k]) < 1e-7, k
    with pytest.raises(NotImplementedError):
        solvers.conelp(L["c"], G, L["h"], kktsolver="ldl", options={"show_progress": False}, **kw)


def test_nt_scaling_at_the_vector_length_of_config_4(golden_dir):
    """G18: the NT-scaling 'l' operations of the reference at ml = 200 000 (BASELINE configs[3]): inputs regenerated from the
    fixture's seed in the generator's order; every 997th entry, the sum and the 2-norm of every output."""
    g = np.load(os.path.join(golden_dir, "g18_nt_scaling_long.npz"))
    ml, st = int(g["ml"]), int(g["stride"])
    rng = np.random.default_rng(int(g["seed"]))
    s = rng.uniform(0.1, 3.0, ml); z = rng.uniform(0.1, 3.0, ml)
    dims = {"l": ml, "q": [], "s": []}

    def check(name, got, tol=1e-14):
        got = np.asarray(got, dtype=float).reshape(-1, order="F")
        assert rel(got[::st], g[name + "_sample"]) < tol, name
        assert abs(got.sum() - float(g[name + "_sum"])) <= 1e-11 * max(1.0, np.abs(got).sum()), name
        assert abs(np.linalg.norm(got) - float(g[name + "_nrm2"])) <= 1e-12 * float(g[name + "_nrm2"]), name
    lm = matrix(0.0, (ml, 1))
    W = misc.compute_scaling(matrix(s), matrix(z), lm, dims)
    check("d", W["d"]._a); check("di", W["di"]._a); check("lmbda", lm._a)
    X = rng.standard_normal((ml, 2))
    for tr in "NT":
        for inv in "NI":
            x = matrix(X.copy(order="F"))
            misc.scale(x, W, trans=tr, inverse=inv)
            check("scale_%s%s" % (tr, inv), x.a)
    x1 = rng.standard_normal(ml); y1 = rng.uniform(0.5, 2.0, ml)
    a = matrix(x1.copy()); misc.scale2(lm, a, dims); check("scale2_N", a._a)
    a = matrix(x1.copy()); misc.scale2(lm, a, dims, inverse="I"); check("scale2_I", a._a)
    a = matrix(x1.copy()); misc.sprod(a, matrix(y1), dims); check("sprod", a._a)
    a = matrix(x1.copy()); misc.sinv(a, matrix(y1), dims); check("sinv", a._a)
    a = matrix(0.0, (ml, 1)); misc.ssqr(a, matrix(x1), dims); check("ssqr", a._a)
    assert abs(misc.sdot(matrix(x1), matrix(y1), dims) - float(g["sdot"])) <= 1e-11 * max(1.0, abs(float(g["sdot"])))
    assert misc.max_step(matrix(x1), dims) == float(g["max_step"])
    ds = rng.uniform(0.2, 2.0, ml); dz = rng.uniform(0.2, 2.0, ml)
    W2 = W_of(W["d"]._a.copy(), W["di"]._a.copy())
    lm2, ms, mz = matrix(lm._a.copy()), matrix(ds.copy()), matrix(dz.copy())
    misc.update_scaling(W2, lm2, ms, mz)
    for got, name in ((ms, "us_s"), (mz, "us_z"), (W2["d"], "us_d"), (W2["di"], "us_di"), (lm2, "us_lmbda")):
        check(name, got._a)


@pytest.mark.parametrize("name", ["grid6x5", "grid6x5_scaled", "eq6x5p4"])
@pytest.mark.parametrize("refinement", [0, 1, 2])
def test_conelp_refinement_golden(golden_dir, name, refinement):
    """G17: conelp with options['refinement'] (coneprog.py:502-507, 599-631, 1211-1235; pure reference, dense-G branch): same
    status, iteration count and solution.  The row-scaled case (rows of G over six decades) takes 32 iterations."""
    g = np.load(os.path.join(golden_dir, "g17_conelp_refinement.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g17_conelp_refinement.json")))["cases"]["%s_r%d" % (name, refinement)]
    P = workloads.lp_grid_eq(6, 5, 4) if name.startswith("eq") else workloads.lp_grid(6, 5)
    gx, hh = P["Gx"], P["h"]
    if name.endswith("scaled"):
        rs = g[name + "_rowscale"]
        gx, hh = gx * rs[P["Gi"]], hh * rs
    G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], gx)
    kw = {"A": spmatrix.from_ccs(P["p"], P["n"], P["Ap"], P["Ai"], P["Ax"]), "b": P["b"]} if "Ap" in P else {}
    sol = lp.conelp(P["c"], G, hh, options={"refinement": refinement}, **kw)
    assert sol["status"] == meta["status"] == "optimal"
    assert sol["iterations"] == meta["iterations"]
    key = "%s_r%d_" % (name, refinement)
    tol = 1e-5 if name.endswith("scaled") else 1e-6
    for k in "xsz":
        assert rel(sol[k], g[key + k]) < tol, k
    assert abs(sol["primal objective"] - meta["primal objective"]) < 1e-7 * max(1.0, abs(meta["primal objective"]))
    with pytest.raises(ValueError):
        lp.conelp(P["c"], G, hh, options={"refinement": -1}, **kw)


@pytest.mark.parametrize("with_eq", [False, True])
def test_refinement_and_coneqp_through_a_user_kktsolver(with_eq):
    """(1) conelp with refinement r solves 1 + 2 (1 + r) systems per iteration (one for the constant part, 1 + r per direction,
    coneprog.py:1211-1235) -- counted through the reference's plug-in point; (2) coneqp / qp take the plug-in too
    (coneprog.py:1969-1981: f solves the system with H = P): same iterates as the default path, 1 factorisation and 2 solves
    per iteration (coneprog.py:2357-2423) after the starting point's one."""
    from kvxopt_amd import solvers
    L = workloads.lp_grid_eq(7, 6, 4) if with_eq else workloads.lp_grid(7, 6)
    Q = workloads.qp_grid(7, 6)
    ml, n = L["ml"], L["n"]
    G = spmatrix.from_ccs(ml, n, L["Gp"], L["Gi"], L["Gx"])
    Gd = np.zeros((ml, n)); Gd[L["Gi"], np.repeat(np.arange(n), np.diff(L["Gp"]))] = L["Gx"]
    p, Ad, kw = 0, np.zeros((0, n)), {}
    if with_eq:
        p = L["p"]
        kw = {"A": spmatrix.from_ccs(p, n, L["Ap"], L["Ai"], L["Ax"]), "b": L["b"]}
        Ad = np.zeros((p, n)); Ad[L["Ai"], np.repeat(np.arange(n), np.diff(L["Ap"]))] = L["Ax"]
    Pd = np.zeros((n, n)); Pd[Q["Pi"], np.repeat(np.arange(n), np.diff(Q["Pp"]))] = Q["Px"]
    Pd = Pd + Pd.T - np.diag(np.diag(Pd))
    calls = {"factor": 0, "solve": 0}

    def dense_kkt(H):
        def factory(W):
            d = np.asarray(W["d"]._a)
            K = np.zeros((n + p + ml, n + p + ml))
            K[:n, :n] = H
            K[:n, n:n + p] = Ad.T; K[:n, n + p:] = Gd.T
            K[n:n + p, :n] = Ad; K[n + p:, :n] = Gd
            K[n + p:, n + p:] = -np.diag(d * d)
            Ki = np.linalg.inv(K)
            calls["factor"] += 1

            def f(x, y, z):
                calls["solve"] += 1
                u = Ki @ np.concatenate([x._a, y._a, z._a])
                x._a[:] = u[:n]
                y._a[:] = u[n:n + p]
                z._a[:] = d * u[n + p:]
            return f
        return factory

    quiet = {"show_progress": False}
    for r in (0, 1, 2):
        ref = solvers.conelp(L["c"], G, L["h"], options=dict(quiet, refinement=r), **kw)
        calls.update(factor=0, solve=0)
        s1 = solvers.conelp(L["c"], G, L["h"], options=dict(quiet, refinement=r), kktsolver=dense_kkt(np.zeros((n, n))), **kw)
        assert s1["status"] == ref["status"] == "optimal" and s1["iterations"] == ref["iterations"]
        assert calls["factor"] == s1["iterations"] + 1
        assert calls["solve"] == 2 + s1["iterations"] * (1 + 2 * (1 + r)), (r, calls)
        for k in ("x", "s", "z"):
            assert rel(s1[k], ref[k]) < 1e-6, (r, k)
    Pm = spmatrix.from_ccs(n, n, Q["Pp"], Q["Pi"], Q["Px"])
    ref = solvers.coneqp(Pm, Q["q"], G, L["h"], options=quiet, **kw)
    calls.update(factor=0, solve=0)
    s2 = solvers.qp(Pm, Q["q"], G, L["h"], options=quiet, kktsolver=dense_kkt(Pd), **kw)
    assert s2["status"] == ref["status"] == "optimal" and s2["iterations"] == ref["iterations"]
    assert calls["factor"] == s2["iterations"] + 1 and calls["solve"] == 1 + 2 * s2["iterations"], calls
    for k in ("x", "s", "z"):
        assert rel(s2[k], ref[k]) < 1e-6, k
    with pytest.raises(NotImplementedError):
        solvers.coneqp(Pm, Q["q"], G, L["h"], options=quiet, kktsolver="ldl", **kw)
