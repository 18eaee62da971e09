"""Generates tests/golden/*.npz|json from the REFERENCE itself (run in the build container only;
/root/reference does not exist on the GPU box, the fixtures it writes travel instead).

What runs:
  * the reference's in-tree C extensions (base, blas, lapack, misc_solvers) built from the sources
    where they lie by oracle/build_ref.sh into oracle/_ref/ (git-ignored), and the reference's own
    src/python/*.py, staged by SYMLINK under /tmp/kvx_ref_stage/kvxopt (never copied into the repo);
  * `kvxopt.cholmod` cannot be built here (SuiteSparse CHOLMOD is absent from the image).  For the
    fixtures that exercise misc.kkt_chol2's sparse branch and conelp with sparse G, the staging
    package gets a TEST DOUBLE for that one module backed by the CPU oracle (oracle/kvx_oracle.c,
    itself pinned on the reference's documented known answers).  Fixtures produced that way are
    tagged "via": "reference python + oracle cholesky"; the pure-reference ones (dense-G LAPACK
    branch, misc_solvers C kernels, base.gemm/syrk) are tagged "via": "reference".

Only numbers are written: inputs and expected outputs.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
STAGE = "/tmp/kvx_ref_stage"
sys.path.insert(0, ROOT)


def stage():
    subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libkvxoracle.so"])
    import shutil
    pkg = os.path.join(STAGE, "kvxopt")
    shutil.rmtree(pkg, ignore_errors=True)
    os.makedirs(pkg)
    for f in os.listdir(os.path.join(ROOT, "oracle", "_ref")):
        if f.endswith(".so"):
            os.symlink(os.path.join(ROOT, "oracle", "_ref", f), os.path.join(pkg, f))
    for f in os.listdir(os.path.join(REF, "src", "python")):
        if f.endswith(".py"):
            os.symlink(os.path.join(REF, "src", "python", f), os.path.join(pkg, f))
    os.symlink(os.path.join(HERE, "ref_cholmod_shim.py"), os.path.join(pkg, "cholmod.py"))
    sys.path.insert(0, STAGE)


def tolist(m):
    return np.array(list(m), dtype=float)


def g1_nt_scaling(kx):
    """NT-scaling 'l' ops of misc.py / misc_solvers.c on seeded inputs."""
    from kvxopt import matrix, misc, misc_solvers
    out = {}
    for ml in (1, 7, 1000):
        rng = np.random.default_rng(100 + ml)
        s = rng.uniform(0.1, 3.0, ml); z = rng.uniform(0.1, 3.0, ml)
        dims = {"l": ml, "q": [], "s": []}
        lm = matrix(0.0, (ml, 1))
        W = misc.compute_scaling(matrix(s), matrix(z), lm, dims)
        case = {"s": s, "z": z, "d": tolist(W["d"]), "di": tolist(W["di"]), "lmbda": tolist(lm)}
        # scale: 1 and 3 columns, all trans/inverse combinations
        X = rng.standard_normal((ml, 3))
        for tr in "NT":
            for inv in "NI":
                x = matrix(X.copy(order="F"))
                misc_solvers.scale(x, W, trans=tr, inverse=inv)
                case["scale_%s%s" % (tr, inv)] = np.array(x)
        case["X"] = X
        x1 = rng.standard_normal(ml); y1 = rng.uniform(0.5, 2.0, ml)
        for name, fn in (("scale2_N", lambda a: misc_solvers.scale2(lm, a, dims)),
                         ("scale2_I", lambda a: misc_solvers.scale2(lm, a, dims, inverse="I")),
                         ("sprod", lambda a: misc_solvers.sprod(a, matrix(y1), dims)),
                         ("sinv", lambda a: misc_solvers.sinv(a, matrix(y1), dims))):
            a = matrix(x1.copy()); fn(a); case[name] = tolist(a)
        a = matrix(0.0, (ml, 1)); misc.ssqr(a, matrix(x1), dims); case["ssqr"] = tolist(a)
        case["x1"] = x1; case["y1"] = y1
        case["sdot"] = np.array(misc_solvers.sdot(matrix(x1), matrix(y1), dims))
        case["max_step"] = np.array(misc_solvers.max_step(matrix(x1), dims))
        # update_scaling (in place)
        ds = rng.uniform(0.2, 2.0, ml); dz = rng.uniform(0.2, 2.0, ml)
        W2 = {"d": matrix(tolist(W["d"])), "di": matrix(tolist(W["di"])), "v": [], "beta": [], "r": [], "rti": []}
        lm2 = matrix(tolist(lm)); ms, mz = matrix(ds.copy()), matrix(dz.copy())
        misc.update_scaling(W2, lm2, ms, mz)
        case.update({"us_ds": ds, "us_dz": dz, "us_s": tolist(ms), "us_z": tolist(mz), "us_d": tolist(W2["d"]),
                     "us_di": tolist(W2["di"]), "us_lmbda": tolist(lm2)})
        for k, v in case.items():
            out["ml%d_%s" % (ml, k)] = v
    np.savez_compressed(os.path.join(HERE, "g1_nt_scaling.npz"), **out)


def ccs(A):
    cp, ri, v = A.CCS
    return np.array(list(cp), dtype=np.int64), np.array(list(ri), dtype=np.int64), np.array(list(v), dtype=float)


def rand_sparse(rng, m, n, dens):
    from kvxopt import spmatrix
    mask = rng.random((m, n)) < dens
    for j in range(n):
        mask[rng.integers(m), j] = True
    for i in range(m):
        if not mask[i].any():
            mask[i, rng.integers(n)] = True
    I, J = np.nonzero(mask)
    V = rng.standard_normal(I.size)
    return spmatrix(V.tolist(), I.tolist(), J.tolist(), (m, n))


def g2_assembly():
    """base.gemm(spdiag(di), G, Gs, partial=True); base.syrk full and partial; S += H."""
    from kvxopt import base, matrix, spmatrix
    rng = np.random.default_rng(202)
    G = rand_sparse(rng, 40, 12, 0.15)
    di = rng.uniform(0.5, 2.0, 40)
    Gs = spmatrix(0.0, G.I, G.J, G.size)
    base.gemm(spmatrix(matrix(di), list(range(40)), list(range(40))), G, Gs, partial=True)
    S = spmatrix([], [], [], (12, 12), "d")
    base.syrk(Gs, S, trans="T")
    out = {}
    out["G_cp"], out["G_ri"], out["G_v"] = ccs(G)
    out["di"] = di
    out["Gs_v"] = ccs(Gs)[2]
    out["S_cp"], out["S_ri"], out["S_v"] = ccs(S)
    di2 = rng.uniform(0.5, 2.0, 40)
    base.gemm(spmatrix(matrix(di2), list(range(40)), list(range(40))), G, Gs, partial=True)
    base.syrk(Gs, S, trans="T", partial=True)
    out["di2"] = di2
    out["S2_v"] = ccs(S)[2]
    H = spmatrix([1.0, 2.0, 0.5, 3.0], [0, 5, 11, 11], [0, 2, 3, 11], (12, 12))
    S += H
    out["H_cp"], out["H_ri"], out["H_v"] = ccs(H)
    out["SH_cp"], out["SH_ri"], out["SH_v"] = ccs(S)
    np.savez_compressed(os.path.join(HERE, "g2_assembly.npz"), **out)


def g3_kkt():
    """misc.kkt_chol2 factor/solve triples: dense-G branch (pure reference, LAPACK) and sparse-G branch
    (reference python + oracle cholesky); p = 0 and p > 0."""
    from kvxopt import matrix, misc, spmatrix
    out = {}
    for tag, p in (("p0", 0), ("p3", 3)):
        rng = np.random.default_rng(303 + p)
        ml, n = 30, 8
        G = rand_sparse(rng, ml, n, 0.25)
        A = rand_sparse(rng, p, n, 0.5) if p else spmatrix([], [], [], (0, n))
        dims = {"l": ml, "q": [], "s": []}
        d = rng.uniform(0.5, 2.0, ml)
        W = {"d": matrix(d), "di": matrix(1.0 / d), "v": [], "beta": [], "r": [], "rti": []}
        bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(ml)
        res = {}
        for branch, Gm, Am in (("dense", matrix(G), matrix(A) if p else matrix(0.0, (0, n))), ("sparse", G, A)):
            f = misc.kkt_chol2(Gm, dims, Am)
            # first call with W = I (as conelp does), then the real W: exercises the fixed-pattern refactor
            W1 = {"d": matrix(1.0, (ml, 1)), "di": matrix(1.0, (ml, 1)), "v": [], "beta": [], "r": [], "rti": []}
            f(W1)
            solve = f(W)
            x, y, z = matrix(bx.copy()), matrix(by.copy()) if p else matrix(0.0, (0, 1)), matrix(bz.copy())
            solve(x, y, z)
            res[branch] = (tolist(x), tolist(y), tolist(z))
        assert np.allclose(res["dense"][0], res["sparse"][0], rtol=1e-9, atol=1e-11), "branches disagree"
        assert np.allclose(res["dense"][2], res["sparse"][2], rtol=1e-9, atol=1e-11)
        out[tag + "_G_cp"], out[tag + "_G_ri"], out[tag + "_G_v"] = ccs(G)
        if p:
            out[tag + "_A_cp"], out[tag + "_A_ri"], out[tag + "_A_v"] = ccs(A)
        out[tag + "_d"] = d
        out[tag + "_bx"], out[tag + "_by"], out[tag + "_bz"] = bx, by, bz
        out[tag + "_x"], out[tag + "_y"], out[tag + "_z"] = res["dense"]          # pure reference result
        out[tag + "_xs"], out[tag + "_ys"], out[tag + "_zs"] = res["sparse"]
    np.savez_compressed(os.path.join(HERE, "g3_kkt_chol2.npz"), **out)


def g4_conelp():
    """conelp traces: structured grid LPs of SURVEY 8(d) config 4b (scaled down) and infeasible cases."""
    from kvxopt import matrix, solvers, spmatrix, misc
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    out = {}
    meta = {}
    cases = [("grid6x5", workloads.lp_grid(6, 5)), ("grid25x20", workloads.lp_grid(25, 20))]
    for name, P in cases:
        ml, n = P["ml"], P["n"]
        cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
        G = spmatrix(P["Gx"].tolist(), P["Gi"].tolist(), cols.tolist(), (ml, n))
        c, h = matrix(P["c"]), matrix(P["h"])
        dvals = []
        dims = {"l": ml, "q": [], "s": []}
        A0 = spmatrix([], [], [], (0, n))
        fac = misc.kkt_chol2(matrix(G), dims, matrix(0.0, (0, n)))

        def kktsolver(W, fac=fac, dvals=dvals):
            dvals.append(tolist(W["d"]))
            return fac(W)
        sol_d = solvers.conelp(c, matrix(G), h, kktsolver=kktsolver)        # pure reference (dense branch)
        sol_s = solvers.conelp(c, G, h)                                     # reference python + oracle cholesky
        assert sol_d["status"] == sol_s["status"] == "optimal"
        assert sol_d["iterations"] == sol_s["iterations"]
        assert np.allclose(tolist(sol_d["x"]), tolist(sol_s["x"]), rtol=1e-7, atol=1e-9)
        out[name + "_x"] = tolist(sol_d["x"]); out[name + "_s"] = tolist(sol_d["s"]); out[name + "_z"] = tolist(sol_d["z"])
        out[name + "_d_per_iter"] = np.array(dvals[1:])       # W['d'] handed to kktsolver at every iteration
        meta[name] = {k: sol_d[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective",
                                           "dual objective", "primal infeasibility", "dual infeasibility")}
        meta[name]["iterations_sparse_branch"] = sol_s["iterations"]
    # primal infeasible: x >= 1 and x <= 0 ;  dual infeasible (unbounded): minimise -x st -x <= 0
    G = spmatrix([-1.0, 1.0], [0, 1], [0, 0], (2, 1))
    sol = solvers.conelp(matrix([1.0]), G, matrix([-1.0, 0.0]))
    meta["primal_infeasible"] = {"status": sol["status"], "iterations": sol["iterations"],
                                 "z": list(sol["z"]), "residual": sol["residual as primal infeasibility certificate"]}
    G = spmatrix([-1.0, -1.0], [0, 1], [0, 1], (2, 2))
    sol = solvers.conelp(matrix([-1.0, 0.5]), G, matrix([0.0, 0.0]))
    meta["dual_infeasible"] = {"status": sol["status"], "iterations": sol["iterations"], "x": list(sol["x"]),
                               "residual": sol["residual as dual infeasibility certificate"]}
    # the documentation LP (examples/doc/chap8/lp.py; tests/test_examples.py:31-34 expects x = [1, 1])
    c = matrix([-4., -5.]); Gd = matrix([[2., 1., -1., 0.], [1., 2., 0., -1.]]); h = matrix([3., 3., 0., 0.])
    from kvxopt import sparse
    sol = solvers.conelp(c, sparse(Gd), h)
    meta["doc_lp"] = {"status": sol["status"], "iterations": sol["iterations"], "x": list(sol["x"]),
                      "primal objective": sol["primal objective"]}
    np.savez_compressed(os.path.join(HERE, "g4_conelp.npz"), **out)
    json.dump({"via": {"*_x/_s/_z/_d_per_iter": "reference (dense-G LAPACK branch), cross-checked against "
                       "reference python + oracle cholesky (sparse branch)",
                       "primal_infeasible/dual_infeasible/doc_lp": "reference python + oracle cholesky"},
               "cases": meta}, open(os.path.join(HERE, "g4_conelp.json"), "w"), indent=1, default=float)


def qp_grid(gx, gy):
    """The config-4b grid LP turned into a convex QP: same G, h (strictly feasible), P = I + (5-point grid
    Laplacian) / 4 (sparse, SPD), q = default_rng(8) standard normal.  Returns the LP dict plus lower-CCS P and q."""
    from kvxopt_amd import workloads
    L = workloads.lp_grid(gx, gy)
    n, cp, ri, vx = workloads.laplacian_2d(gx, gy)
    assert n == L["n"]
    Px = vx * 0.25
    Px[cp[:-1]] += 1.0                                  # the diagonal is the first entry of every lower column
    L.update({"Pp": cp, "Pi": ri, "Px": Px, "q": np.random.default_rng(8).standard_normal(n)})
    return L


def g5_coneqp():
    """coneqp traces on the grid QPs (misc.kkt_chol2 with H = P: S = P + G' W^-1 W^-T G, misc.py:1425-1426, 1454-1455)."""
    from kvxopt import matrix, solvers, spmatrix, misc
    solvers.options["show_progress"] = False
    out, meta = {}, {}
    for name, gx, gy in [("qp6x5", 6, 5), ("qp25x20", 25, 20)]:
        Q = qp_grid(gx, gy)
        ml, n = Q["ml"], Q["n"]
        cols = np.repeat(np.arange(n), np.diff(Q["Gp"]))
        G = spmatrix(Q["Gx"].tolist(), Q["Gi"].tolist(), cols.tolist(), (ml, n))
        pcols = np.repeat(np.arange(n), np.diff(Q["Pp"]))
        P = spmatrix(Q["Px"].tolist(), Q["Pi"].tolist(), pcols.tolist(), (n, n))      # lower triangle ('L' storage, coneprog.py:1452-1455)
        q, h = matrix(Q["q"]), matrix(Q["h"])
        dims = {"l": ml, "q": [], "s": []}
        dvals = []
        fac = misc.kkt_chol2(matrix(G), dims, matrix(0.0, (0, n)))

        def kktsolver(W, fac=fac, dvals=dvals, P=P):
            dvals.append(tolist(W["d"]))
            return fac(W, P)
        sol_d = solvers.coneqp(P, q, matrix(G), h, kktsolver=kktsolver)      # pure reference (dense-G LAPACK branch)
        sol_s = solvers.coneqp(P, q, G, h)                                    # reference python + oracle cholesky
        assert sol_d["status"] == sol_s["status"] == "optimal"
        assert sol_d["iterations"] == sol_s["iterations"]
        assert np.allclose(tolist(sol_d["x"]), tolist(sol_s["x"]), rtol=1e-7, atol=1e-9)
        out[name + "_x"] = tolist(sol_d["x"]); out[name + "_s"] = tolist(sol_d["s"]); out[name + "_z"] = tolist(sol_d["z"])
        out[name + "_d_per_iter"] = np.array(dvals[1:])
        meta[name] = {k: sol_d[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective",
                                           "dual objective", "primal infeasibility", "dual infeasibility",
                                           "primal slack", "dual slack")}
    np.savez_compressed(os.path.join(HERE, "g5_coneqp.npz"), **out)
    json.dump({"via": "reference (dense-G LAPACK branch), cross-checked against reference python + oracle cholesky "
                      "(sparse branch); problem generator: make_goldens.qp_grid (mirrored in kvxopt_amd.workloads.qp_grid)",
               "cases": meta}, open(os.path.join(HERE, "g5_coneqp.json"), "w"), indent=1, default=float)


def g6_conelp_std():
    """conelp on the standard-form grid LP of SURVEY 8(d) config 4a (equality constraints, G = -I): the
    misc.kkt_chol2 branch with K = A S^-1 A' (misc.py:1483-1487, 1545)."""
    from kvxopt import matrix, solvers, spmatrix, misc
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    out, meta = {}, {}
    for name, gx, gy in [("std6x5", 6, 5), ("std15x12", 15, 12)]:
        L = workloads.lp_grid_std(gx, gy)
        p, nv = L["p"], L["n"]
        acols = np.repeat(np.arange(nv), np.diff(L["Ap"]))
        A = spmatrix(L["Ax"].tolist(), L["Ai"].tolist(), acols.tolist(), (p, nv))
        G = spmatrix((-np.ones(nv)).tolist(), list(range(nv)), list(range(nv)), (nv, nv))
        c, h, b = matrix(L["c"]), matrix(L["h"]), matrix(L["b"])
        sol_d = solvers.conelp(c, matrix(G), h, A=matrix(A), b=b, kktsolver="chol2")   # pure reference (dense branch)
        sol_s = solvers.conelp(c, G, h, A=A, b=b)                                      # reference python + oracle cholesky
        assert sol_d["status"] == sol_s["status"] == "optimal", (sol_d["status"], sol_s["status"])
        assert sol_d["iterations"] == sol_s["iterations"]
        assert np.allclose(tolist(sol_d["x"]), tolist(sol_s["x"]), rtol=1e-6, atol=1e-8)
        for k in "xysz":
            out[name + "_" + k] = tolist(sol_d[k])
        meta[name] = {k: sol_d[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective",
                                           "dual objective", "primal infeasibility", "dual infeasibility")}
    np.savez_compressed(os.path.join(HERE, "g6_conelp_std.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch of misc.kkt_chol2), cross-checked against reference python + "
                      "oracle cholesky (sparse branch); generator kvxopt_amd.workloads.lp_grid_std",
               "cases": meta}, open(os.path.join(HERE, "g6_conelp_std.json"), "w"), indent=1, default=float)


def g8_conelp_eq():
    """conelp with a general sparse G AND equality constraints (workloads.lp_grid_eq): the branch of misc.kkt_chol2 that
    forms K = A S^-1 A' with a non-diagonal S (misc.py:1476-1487, 1545)."""
    from kvxopt import matrix, solvers, spmatrix
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    out, meta = {}, {}
    for name, gx, gy, p in [("eq6x5p4", 6, 5, 4), ("eq15x12p20", 15, 12, 20)]:
        L = workloads.lp_grid_eq(gx, gy, p)
        ml, n = L["ml"], L["n"]
        gcols = np.repeat(np.arange(n), np.diff(L["Gp"]))
        acols = np.repeat(np.arange(n), np.diff(L["Ap"]))
        G = spmatrix(L["Gx"].tolist(), L["Gi"].tolist(), gcols.tolist(), (ml, n))
        A = spmatrix(L["Ax"].tolist(), L["Ai"].tolist(), acols.tolist(), (p, n))
        c, h, b = matrix(L["c"]), matrix(L["h"]), matrix(L["b"])
        sol_d = solvers.conelp(c, matrix(G), h, A=matrix(A), b=b, kktsolver="chol2")   # pure reference (dense LAPACK branch)
        sol_s = solvers.conelp(c, G, h, A=A, b=b)                                      # reference python + oracle cholesky
        assert sol_d["status"] == sol_s["status"] == "optimal", (sol_d["status"], sol_s["status"])
        assert sol_d["iterations"] == sol_s["iterations"]
        assert np.allclose(tolist(sol_d["x"]), tolist(sol_s["x"]), rtol=1e-6, atol=1e-8)
        for k in "xysz":
            out[name + "_" + k] = tolist(sol_d[k])
        meta[name] = {k: sol_d[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective",
                                           "dual objective", "primal infeasibility", "dual infeasibility")}
    np.savez_compressed(os.path.join(HERE, "g8_conelp_eq.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch of misc.kkt_chol2), cross-checked against reference python + "
                      "oracle cholesky (sparse branch); generator kvxopt_amd.workloads.lp_grid_eq",
               "cases": meta}, open(os.path.join(HERE, "g8_conelp_eq.json"), "w"), indent=1, default=float)


def g9_coneqp_eq():
    """coneqp with equality constraints: the grid QP of G5 plus the equality rows of G8 (misc.kkt_chol2 with H = P and
    K = A S^-1 A', misc.py:1425-1426, 1476-1487)."""
    from kvxopt import matrix, solvers, spmatrix
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    out, meta = {}, {}
    for name, gx, gy, p in [("qpeq6x5p4", 6, 5, 4), ("qpeq15x12p20", 15, 12, 20)]:
        Q = qp_grid(gx, gy)
        L = workloads.lp_grid_eq(gx, gy, p)
        ml, n = Q["ml"], Q["n"]
        cols = np.repeat(np.arange(n), np.diff(Q["Gp"]))
        G = spmatrix(Q["Gx"].tolist(), Q["Gi"].tolist(), cols.tolist(), (ml, n))
        pcols = np.repeat(np.arange(n), np.diff(Q["Pp"]))
        P = spmatrix(Q["Px"].tolist(), Q["Pi"].tolist(), pcols.tolist(), (n, n))
        acols = np.repeat(np.arange(n), np.diff(L["Ap"]))
        A = spmatrix(L["Ax"].tolist(), L["Ai"].tolist(), acols.tolist(), (p, n))
        q, h, b = matrix(Q["q"]), matrix(Q["h"]), matrix(L["b"])
        sol_d = solvers.coneqp(P, q, matrix(G), h, A=matrix(A), b=b, kktsolver="chol2")   # pure reference (dense LAPACK branch)
        sol_s = solvers.coneqp(P, q, G, h, A=A, b=b)                                      # reference python + oracle cholesky
        assert sol_d["status"] == sol_s["status"] == "optimal", (sol_d["status"], sol_s["status"])
        assert sol_d["iterations"] == sol_s["iterations"]
        assert np.allclose(tolist(sol_d["x"]), tolist(sol_s["x"]), rtol=1e-6, atol=1e-8)
        for k in "xysz":
            out[name + "_" + k] = tolist(sol_d[k])
        meta[name] = {k: sol_d[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective",
                                           "dual objective", "primal infeasibility", "dual infeasibility")}
    np.savez_compressed(os.path.join(HERE, "g9_coneqp_eq.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch of misc.kkt_chol2), cross-checked against reference python + "
                      "oracle cholesky (sparse branch); generators make_goldens.qp_grid + kvxopt_amd.workloads.lp_grid_eq",
               "cases": meta}, open(os.path.join(HERE, "g9_coneqp_eq.json"), "w"), indent=1, default=float)


def g10_conelp_starts():
    """conelp with user-supplied starting points (coneprog.py:683-737, 806-842): the strictly feasible primal / dual
    points the grid-LP generator is built from, all three combinations; pure reference (dense LAPACK branch)."""
    from kvxopt import matrix, solvers, spmatrix
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    P = workloads.lp_grid(15, 12)
    ml, n = P["ml"], P["n"]
    cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
    G = matrix(spmatrix(P["Gx"].tolist(), P["Gi"].tolist(), cols.tolist(), (ml, n)))
    c, h = matrix(P["c"]), matrix(P["h"])
    ps = {"x": matrix(P["x0"]), "s": matrix(P["s0"])}
    ds = {"z": matrix(P["z0"])}
    out, meta = {}, {}
    for name, kw in (("primal", {"primalstart": ps}), ("dual", {"dualstart": ds}), ("both", {"primalstart": ps, "dualstart": ds})):
        sol = solvers.conelp(c, G, h, kktsolver="chol2", **kw)
        assert sol["status"] == "optimal"
        for k in "xsz":
            out[name + "_" + k] = tolist(sol[k])
        meta[name] = {k: sol[k] for k in ("status", "iterations", "gap", "primal objective", "dual objective")}
    np.savez_compressed(os.path.join(HERE, "g10_conelp_starts.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch); generator kvxopt_amd.workloads.lp_grid(15, 12) with its x0, s0, z0",
               "cases": meta}, open(os.path.join(HERE, "g10_conelp_starts.json"), "w"), indent=1, default=float)


def g11_coneqp_initvals():
    """coneqp with user-supplied initial values (coneprog.py:2108-2150) on the grid QP: all four given, and x / s only."""
    from kvxopt import matrix, solvers, spmatrix
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    Q = qp_grid(15, 12)
    L = workloads.lp_grid(15, 12)
    ml, n = Q["ml"], Q["n"]
    cols = np.repeat(np.arange(n), np.diff(Q["Gp"]))
    G = matrix(spmatrix(Q["Gx"].tolist(), Q["Gi"].tolist(), cols.tolist(), (ml, n)))
    pcols = np.repeat(np.arange(n), np.diff(Q["Pp"]))
    P = spmatrix(Q["Px"].tolist(), Q["Pi"].tolist(), pcols.tolist(), (n, n))
    q, h = matrix(Q["q"]), matrix(Q["h"])
    out, meta = {}, {}
    for name, iv in (("all", {"x": matrix(L["x0"]), "s": matrix(L["s0"]), "z": matrix(L["z0"])}),
                     ("xs", {"x": matrix(L["x0"]), "s": matrix(L["s0"])})):
        sol = solvers.coneqp(P, q, G, h, initvals=iv, kktsolver="chol2")
        assert sol["status"] == "optimal"
        for k in "xsz":
            out[name + "_" + k] = tolist(sol[k])
        meta[name] = {k: sol[k] for k in ("status", "iterations", "gap", "primal objective", "dual objective")}
    np.savez_compressed(os.path.join(HERE, "g11_coneqp_initvals.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch); generators make_goldens.qp_grid(15, 12), workloads.lp_grid(15, 12) x0, s0, z0",
               "cases": meta}, open(os.path.join(HERE, "g11_coneqp_initvals.json"), "w"), indent=1, default=float)


def g7_mps():
    """modeling.op.fromfile + solve on the reference's own fixture tests/boeing2.mps (tests/test_modeling.py:59-63):
    the parsed problem (objective, inequality / equality counts) and the solution of the reference's default LP path
    (solvers.lp -> conelp, dense LAPACK kkt 'chol': pure reference).  The MPS file itself is DATA (Netlib LP) and is
    kept next to the goldens as tests/golden/boeing2.mps."""
    import shutil
    from kvxopt import solvers
    from kvxopt.modeling import op
    solvers.options["show_progress"] = False
    src = os.path.join(REF, "tests", "boeing2.mps")
    shutil.copyfile(src, os.path.join(HERE, "boeing2.mps"))
    lp = op()
    lp.fromfile(src)
    lp.solve()
    assert lp.status == "optimal"
    xs = {v.name: float(v.value[0]) for v in lp.variables()}
    meta = {"via": "reference (modeling.op.fromfile, solvers.lp with the dense kkt solver)", "status": lp.status,
            "objective": float(lp.objective.value()[0]), "n_variables": len(lp.variables()),
            "n_inequalities": len(lp.inequalities()), "n_equalities": len(lp.equalities()), "x": xs}
    json.dump(meta, open(os.path.join(HERE, "g7_boeing2.json"), "w"), indent=1)
    print("boeing2: objective", meta["objective"], "vars", meta["n_variables"], "ineq", meta["n_inequalities"], "eq", meta["n_equalities"])


def g12_nonlinear_block():
    """The nonlinear block of cvxprog on the kkt_chol2 path (mnl > 0): compute_scaling / update_scaling / scale with
    W['dnl'], and misc.kkt_chol2(G, dims, A, mnl)(W, H, Df) -- dense branch (pure reference, LAPACK) and sparse branch
    (reference python + oracle cholesky); two factor calls (the second refactors on the fixed pattern with a new Df)."""
    from kvxopt import matrix, misc, spmatrix
    out = {}
    mnl, ml, n = 5, 30, 9
    dims = {"l": ml, "q": [], "s": []}
    rng = np.random.default_rng(1212)
    s, z = rng.uniform(0.1, 3.0, mnl + ml), rng.uniform(0.1, 3.0, mnl + ml)
    lm = matrix(0.0, (mnl + ml, 1))
    W = misc.compute_scaling(matrix(s), matrix(z), lm, dims, mnl)
    out["s"], out["z"] = s, z
    for k in ("dnl", "dnli", "d", "di"):
        out["cs_" + k] = tolist(W[k])
    out["cs_lmbda"] = tolist(lm)
    xs = rng.standard_normal((mnl + ml, 3))
    for inv in ("N", "I"):
        X = matrix(xs.copy())
        misc.scale(X, W, trans="T", inverse=inv)
        out["scale_" + inv] = np.array(X).reshape(mnl + ml, 3)
    out["scale_in"] = xs
    s2, z2 = rng.uniform(0.5, 2.0, mnl + ml), rng.uniform(0.5, 2.0, mnl + ml)
    ms, mz = matrix(s2.copy()), matrix(z2.copy())
    misc.update_scaling(W, lm, ms, mz)
    out["us_s_in"], out["us_z_in"] = s2, z2
    out["us_s"], out["us_z"], out["us_lmbda"] = tolist(ms), tolist(mz), tolist(lm)
    for k in ("dnl", "dnli", "d", "di"):
        out["us_" + k] = tolist(W[k])
    for tag, p in (("p0", 0), ("p2", 2)):
        rng = np.random.default_rng(1213 + p)
        G = rand_sparse(rng, ml, n, 0.25)
        Df1 = rand_sparse(rng, mnl, n, 0.5)
        Df2 = spmatrix(rng.standard_normal(len(Df1)).tolist(), Df1.I, Df1.J, Df1.size)
        A = rand_sparse(rng, p, n, 0.5) if p else spmatrix([], [], [], (0, n))
        hI = list(range(n)) + [3, 7]
        hJ = list(range(n)) + [1, 2]
        H1 = spmatrix(rng.uniform(0.5, 1.5, n).tolist() + [0.2, -0.1], hI, hJ, (n, n))
        H2 = spmatrix(rng.uniform(0.5, 1.5, n).tolist() + [-0.3, 0.15], hI, hJ, (n, n))
        Ws = []
        for _ in range(2):
            d, dnl = rng.uniform(0.5, 2.0, ml), rng.uniform(0.5, 2.0, mnl)
            Ws.append({"dnl": matrix(dnl), "dnli": matrix(1.0 / dnl), "d": matrix(d), "di": matrix(1.0 / d),
                       "v": [], "beta": [], "r": [], "rti": []})
        bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(mnl + ml)
        res = {}
        for branch in ("dense", "sparse"):
            dn = branch == "dense"
            cv = (lambda M: matrix(M)) if dn else (lambda M: M)
            Am = (matrix(A) if p else matrix(0.0, (0, n))) if dn else A
            f = misc.kkt_chol2(cv(G), dims, Am, mnl)
            f(Ws[0], cv(H1), cv(Df1))
            solve = f(Ws[1], cv(H2), cv(Df2))
            x, y, zz = matrix(bx.copy()), matrix(by.copy()) if p else matrix(0.0, (0, 1)), matrix(bz.copy())
            solve(x, y, zz)
            res[branch] = (tolist(x), tolist(y), tolist(zz))
        for a, b in zip(res["dense"], res["sparse"]):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-11), "branches disagree"
        out[tag + "_G_cp"], out[tag + "_G_ri"], out[tag + "_G_v"] = ccs(G)
        out[tag + "_Df_cp"], out[tag + "_Df_ri"], out[tag + "_Df1_v"] = ccs(Df1)
        out[tag + "_Df2_v"] = ccs(Df2)[2]
        out[tag + "_H_cp"], out[tag + "_H_ri"], out[tag + "_H1_v"] = ccs(H1)
        out[tag + "_H2_v"] = ccs(H2)[2]
        if p:
            out[tag + "_A_cp"], out[tag + "_A_ri"], out[tag + "_A_v"] = ccs(A)
        for i in range(2):
            for k in ("dnl", "d"):
                out["%s_W%d_%s" % (tag, i, k)] = tolist(Ws[i][k])
        out[tag + "_bx"], out[tag + "_by"], out[tag + "_bz"] = bx, by, bz
        out[tag + "_x"], out[tag + "_y"], out[tag + "_z"] = res["dense"]          # pure reference result
    np.savez_compressed(os.path.join(HERE, "g12_nonlinear_block.npz"), **out)


def g14_kkt_singular():
    """misc.kkt_chol2 when S = G' W^-2 G is singular (G has a zero column) and A restores full rank: the reference's
    fallback S + A'A (misc.py:1433-1447, 1525-1526), dense-G LAPACK branch = pure reference.  Also a p > 0 case with a
    G that has ONE entry per row (diagonal S: standard-form LPs)."""
    from kvxopt import matrix, misc, spmatrix
    out = {}
    rng = np.random.default_rng(1400)
    ml, n, p = 14, 6, 2
    Gd = rng.standard_normal((ml, n)) * (rng.uniform(size=(ml, n)) < 0.5)
    Gd[:, 4] = 0.0                                        # column 4 never appears in an inequality
    for j in range(n):
        if j != 4 and not Gd[:, j].any():
            Gd[j, j] = 1.0
    Ad = rng.standard_normal((p, n)) * (rng.uniform(size=(p, n)) < 0.7)
    Ad[0, 4] = 1.5; Ad[1, 0] = -0.5
    I, J = np.nonzero(Gd); G = spmatrix(Gd[I, J].tolist(), I.tolist(), J.tolist(), (ml, n))
    I, J = np.nonzero(Ad); A = spmatrix(Ad[I, J].tolist(), I.tolist(), J.tolist(), (p, n))
    dims = {"l": ml, "q": [], "s": []}
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(ml)
    f = misc.kkt_chol2(matrix(G), dims, matrix(A))
    sols = []
    ds = [np.ones(ml), rng.uniform(0.5, 2.0, ml)]
    for d in ds:
        W = {"d": matrix(d), "di": matrix(1.0 / d), "v": [], "beta": [], "r": [], "rti": []}
        solve = f(W)
        x, y, z = matrix(bx.copy()), matrix(by.copy()), matrix(bz.copy())
        solve(x, y, z)
        sols.append((tolist(x), tolist(y), tolist(z)))
    out["sing_G_cp"], out["sing_G_ri"], out["sing_G_v"] = ccs(G)
    out["sing_A_cp"], out["sing_A_ri"], out["sing_A_v"] = ccs(A)
    out["sing_d"] = ds[1]
    out["sing_bx"], out["sing_by"], out["sing_bz"] = bx, by, bz
    for i, (x, y, z) in enumerate(sols):
        out["sing_x%d" % i], out["sing_y%d" % i], out["sing_z%d" % i] = x, y, z
    # diagonal S with equality rows (G = -I stacked twice with different signs: one entry per row)
    n2, p2 = 7, 3
    ml2 = 2 * n2
    G2 = spmatrix([-1.0] * n2 + [0.5] * n2, list(range(ml2)), list(range(n2)) * 2, (ml2, n2))
    A2d = rng.standard_normal((p2, n2)) * (rng.uniform(size=(p2, n2)) < 0.6)
    for i in range(p2):
        A2d[i, i] = 1.0 + i
    I, J = np.nonzero(A2d); A2 = spmatrix(A2d[I, J].tolist(), I.tolist(), J.tolist(), (p2, n2))
    d2 = rng.uniform(0.5, 2.0, ml2)
    bx2, by2, bz2 = rng.standard_normal(n2), rng.standard_normal(p2), rng.standard_normal(ml2)
    f2 = misc.kkt_chol2(matrix(G2), {"l": ml2, "q": [], "s": []}, matrix(A2))
    f2({"d": matrix(1.0, (ml2, 1)), "di": matrix(1.0, (ml2, 1)), "v": [], "beta": [], "r": [], "rti": []})
    solve = f2({"d": matrix(d2), "di": matrix(1.0 / d2), "v": [], "beta": [], "r": [], "rti": []})
    x, y, z = matrix(bx2.copy()), matrix(by2.copy()), matrix(bz2.copy())
    solve(x, y, z)
    out["diag_G_cp"], out["diag_G_ri"], out["diag_G_v"] = ccs(G2)
    out["diag_A_cp"], out["diag_A_ri"], out["diag_A_v"] = ccs(A2)
    out["diag_d"] = d2
    out["diag_bx"], out["diag_by"], out["diag_bz"] = bx2, by2, bz2
    out["diag_x"], out["diag_y"], out["diag_z"] = tolist(x), tolist(y), tolist(z)
    np.savez_compressed(os.path.join(HERE, "g14_kkt_singular.npz"), **out)


def g15_q_cone_scaling():
    """Nesterov-Todd scaling with second-order-cone blocks (dims = {'l': 4, 'q': [5, 1, 9], 's': []}, also with a nonlinear
    block): misc.compute_scaling / update_scaling / ssqr and misc_solvers.scale / scale2 / sprod / sinv / sdot / max_step on
    seeded interior points -- pure reference."""
    from kvxopt import matrix, misc, misc_solvers
    out = {}
    for tag, mnl in (("a", None), ("b", 3)):
        rng = np.random.default_rng(1500 + (mnl or 0))
        k = mnl or 0
        ml, q = 4, [5, 1, 9]
        dims = {"l": ml, "q": q, "s": []}
        N = k + ml + sum(q)

        def interior():
            x = rng.uniform(0.3, 2.0, N)
            ind = k + ml
            for m in q:
                t = rng.standard_normal(m - 1)
                x[ind + 1:ind + m] = t
                x[ind] = np.linalg.norm(t) + rng.uniform(0.2, 1.5)
                ind += m
            return x
        s, z = interior(), interior()
        lm = matrix(0.0, (N, 1))
        W = misc.compute_scaling(matrix(s), matrix(z), lm, dims, mnl)
        out[tag + "_s"], out[tag + "_z"] = s, z
        out[tag + "_lmbda"] = tolist(lm)
        for key in ("d", "di") + (("dnl", "dnli") if mnl else ()):
            out[tag + "_" + key] = tolist(W[key])
        out[tag + "_beta"] = np.array(W["beta"])
        out[tag + "_v"] = np.concatenate([tolist(v) for v in W["v"]])
        X = rng.standard_normal((N, 2))
        out[tag + "_X"] = X
        for tr in "NT":
            for inv in "NI":
                x = matrix(X.copy(order="F"))
                misc_solvers.scale(x, W, trans=tr, inverse=inv)
                out["%s_scale_%s%s" % (tag, tr, inv)] = np.array(x)
        x1, y1 = interior() * 0.7 + 0.1 * rng.standard_normal(N), interior()
        out[tag + "_x1"], out[tag + "_y1"] = x1, y1
        for name, fn in (("scale2_N", lambda a: misc_solvers.scale2(lm, a, dims, k)),
                         ("scale2_I", lambda a: misc_solvers.scale2(lm, a, dims, k, inverse="I")),
                         ("sprod", lambda a: misc_solvers.sprod(a, matrix(y1), dims, k)),
                         ("sinv", lambda a: misc_solvers.sinv(a, matrix(y1), dims, k))):
            a = matrix(x1.copy()); fn(a); out[tag + "_" + name] = tolist(a)
        a = matrix(0.0, (N, 1)); misc.ssqr(a, matrix(y1), dims, k); out[tag + "_ssqr"] = tolist(a)
        out[tag + "_sdot"] = np.array(misc_solvers.sdot(matrix(x1), matrix(y1), dims, k))
        out[tag + "_max_step"] = np.array(misc_solvers.max_step(matrix(x1), dims, k))
        # update_scaling: new iterates in the current scaling, close to lmbda (as the interior-point update produces them)
        ns = np.array(lm).ravel() * (1.0 + 0.05 * rng.standard_normal(N))
        nz = np.array(lm).ravel() * (1.0 + 0.05 * rng.standard_normal(N))
        ms, mz = matrix(ns.copy()), matrix(nz.copy())
        out[tag + "_us_s_in"], out[tag + "_us_z_in"] = ns, nz
        misc.update_scaling(W, lm, ms, mz)
        out[tag + "_us_s"], out[tag + "_us_z"], out[tag + "_us_lmbda"] = tolist(ms), tolist(mz), tolist(lm)
        for key in ("d", "di") + (("dnl", "dnli") if mnl else ()):
            out[tag + "_us_" + key] = tolist(W[key])
        out[tag + "_us_beta"] = np.array(W["beta"])
        out[tag + "_us_v"] = np.concatenate([tolist(v) for v in W["v"]])
    np.savez(os.path.join(HERE, "g15_q_cone_scaling.npz"), **out)


def g16_s_cone_scaling():
    """Nesterov-Todd scaling with semidefinite blocks (dims = {'l': 3, 'q': [4], 's': [3, 1, 6]}, also with a nonlinear block):
    misc.compute_scaling / update_scaling / ssqr, misc_solvers.scale / scale2 / sprod (both forms) / sinv / sdot / max_step (with
    and without sigma) and the storage helpers pack / pack2 / unpack / symm / trisc / triusc on seeded interior points -- pure
    reference (LAPACK potrf / gesvd / syevd / syevr of the OpenBLAS that ships in scipy)."""
    from kvxopt import matrix, misc, misc_solvers
    out = {}
    for tag, mnl in (("a", None), ("b", 2)):
        rng = np.random.default_rng(1600 + (mnl or 0))
        k = mnl or 0
        ml, q, sd = 3, [4], [3, 1, 6]
        dims = {"l": ml, "q": q, "s": sd}
        nlq = k + ml + sum(q)
        N = nlq + sum(m * m for m in sd)
        Nd = nlq + sum(sd)                                   # length with diagonal 's' storage

        def spd(m, spread=1.0):
            B = rng.standard_normal((m, m))
            return B @ B.T / m + np.diag(rng.uniform(0.3, 0.3 + spread, m))

        def interior():
            x = rng.uniform(0.3, 2.0, N)
            ind = k + ml
            for m in q:
                t = rng.standard_normal(m - 1)
                x[ind + 1:ind + m] = t
                x[ind] = np.linalg.norm(t) + rng.uniform(0.2, 1.5)
                ind += m
            for m in sd:
                x[ind:ind + m * m] = spd(m).reshape(-1, order="F")
                ind += m * m
            return x
        s, z = interior(), interior()
        lm = matrix(0.0, (Nd, 1))
        W = misc.compute_scaling(matrix(s), matrix(z), lm, dims, mnl)
        out[tag + "_s"], out[tag + "_z"] = s, z
        out[tag + "_lmbda"] = tolist(lm)
        out[tag + "_r"] = np.concatenate([tolist(r) for r in W["r"]])
        out[tag + "_rti"] = np.concatenate([tolist(r) for r in W["rti"]])
        X = rng.standard_normal((N, 2))                      # 's' blocks: nonsymmetric on purpose (only the lower triangle is read)
        out[tag + "_X"] = X
        for tr in "NT":
            for inv in "NI":
                x = matrix(X.copy(order="F"))
                misc_solvers.scale(x, W, trans=tr, inverse=inv)
                out["%s_scale_%s%s" % (tag, tr, inv)] = np.array(x)
        x1 = interior() * 0.7 + 0.1 * rng.standard_normal(N)
        y1 = interior()
        out[tag + "_x1"], out[tag + "_y1"] = x1, y1
        for name, inv in (("scale2_N", "N"), ("scale2_I", "I")):
            a = matrix(x1.copy()); misc_solvers.scale2(lm, a, dims, k, inverse=inv); out[tag + "_" + name] = tolist(a)
        a, b = matrix(x1.copy()), matrix(y1.copy())
        misc_solvers.sprod(a, b, dims, k)
        out[tag + "_sprod"], out[tag + "_sprod_y_after"] = tolist(a), tolist(b)
        yd = np.concatenate([y1[:nlq], rng.uniform(0.4, 2.0, sum(sd))])       # diagonal 's' storage
        out[tag + "_yd"] = yd
        a = matrix(x1.copy()); misc_solvers.sprod(a, matrix(yd), dims, k, diag="D"); out[tag + "_sprod_D"] = tolist(a)
        a = matrix(x1.copy()); misc_solvers.sinv(a, matrix(yd), dims, k); out[tag + "_sinv"] = tolist(a)
        a = matrix(0.0, (Nd, 1)); misc.ssqr(a, matrix(yd), dims, k); out[tag + "_ssqr"] = tolist(a)
        out[tag + "_sdot"] = np.array(misc_solvers.sdot(matrix(x1), matrix(y1), dims, k))
        out[tag + "_max_step"] = np.array(misc_solvers.max_step(matrix(x1), dims, k))
        xs, sg = matrix(x1.copy()), matrix(0.0, (sum(sd), 1))
        out[tag + "_max_step_sigma_t"] = np.array(misc_solvers.max_step(xs, dims, k, sg))
        out[tag + "_max_step_sigma"], out[tag + "_max_step_x"] = tolist(sg), tolist(xs)
        # storage helpers
        npk = nlq + sum(m * (m + 1) // 2 for m in sd)
        yp = matrix(0.0, (npk + 3, 1)); misc_solvers.pack(matrix(x1), yp, dims, k, 0, 2); out[tag + "_pack"] = tolist(yp)
        yu = matrix(7.0, (N + 1, 1)); misc_solvers.unpack(yp, yu, dims, k, 2, 1); out[tag + "_unpack"] = tolist(yu)
        x2 = matrix(np.column_stack([x1, y1]).copy(order="F")); misc_solvers.pack2(x2, dims, k); out[tag + "_pack2"] = np.array(x2)
        if mnl is None:
            a = matrix(x1.copy()); misc_solvers.trisc(a, dims); out[tag + "_trisc"] = tolist(a)
            a = matrix(x1.copy()); misc_solvers.triusc(a, dims); out[tag + "_triusc"] = tolist(a)
            a = matrix(x1.copy()); misc_solvers.symm(a, 6, nlq + 10); out[tag + "_symm"] = tolist(a)
        # update_scaling: new iterates in the current scaling, close to lmbda; their 's' components arrive as Cholesky factors
        lmv = np.array(lm).ravel()
        ns_ = np.zeros(N); nz_ = np.zeros(N)
        ns_[:nlq] = lmv[:nlq] * (1.0 + 0.05 * rng.standard_normal(nlq))
        nz_[:nlq] = lmv[:nlq] * (1.0 + 0.05 * rng.standard_normal(nlq))
        ind, il = nlq, nlq
        for m in sd:
            for vec in (ns_, nz_):
                E = 0.05 * rng.standard_normal((m, m))
                M = np.diag(lmv[il:il + m]) + 0.5 * (E + E.T)
                vec[ind:ind + m * m] = np.linalg.cholesky(M).reshape(-1, order="F")
            ind += m * m
            il += m
        ms, mz = matrix(ns_.copy()), matrix(nz_.copy())
        out[tag + "_us_s_in"], out[tag + "_us_z_in"] = ns_, nz_
        misc.update_scaling(W, lm, ms, mz)
        out[tag + "_us_s"], out[tag + "_us_z"], out[tag + "_us_lmbda"] = tolist(ms), tolist(mz), tolist(lm)
        out[tag + "_us_r"] = np.concatenate([tolist(r) for r in W["r"]])
        out[tag + "_us_rti"] = np.concatenate([tolist(r) for r in W["rti"]])
    np.savez(os.path.join(HERE, "g16_s_cone_scaling.npz"), **out)


def g13_gemv_subblocks():
    """base.gemv (base.c:744-851 -> sparse.c:1073-1104) with the sub-block arguments m, n, offsetA and strides, sparse A."""
    from kvxopt import base, matrix, spmatrix
    rng = np.random.default_rng(1300)
    A, (cp, ri, v) = None, (None, None, None)
    M, N = 9, 7
    D = rng.standard_normal((M, N)) * (rng.uniform(size=(M, N)) < 0.45)
    I, J = np.nonzero(D)
    A = spmatrix(D[I, J].tolist(), I.tolist(), J.tolist(), (M, N))
    cp, ri, v = ccs(A)
    out = {"M": np.array(M), "N": np.array(N), "cp": cp, "ri": ri, "v": v}
    cases = []
    for ci, (trans, m, n, oi, oj, incx, incy, ox, oy, alpha, beta) in enumerate([
            ("N", 4, 3, 2, 1, 1, 1, 0, 0, 1.0, 0.0), ("T", 4, 3, 2, 1, 1, 1, 0, 0, -2.0, 0.5),
            ("N", 9, 7, 0, 0, 2, 1, 1, 3, 0.5, -1.0), ("T", 5, 5, 4, 2, 1, 2, 2, 0, 1.5, 1.0),
            ("N", 3, 2, 6, 5, -1, 1, 0, 1, 1.0, 2.0), ("T", 6, 4, 1, 3, 1, -2, 0, 0, -1.0, 0.25),
            ("N", 0, 3, 0, 0, 1, 1, 0, 0, 1.0, 0.0), ("N", 4, 0, 1, 1, 1, 1, 0, 0, 1.0, 0.5)]):
        lx, ly = (n, m) if trans == "N" else (m, n)
        x = rng.standard_normal(ox + max(lx - 1, 0) * abs(incx) + 3)
        y = rng.standard_normal(oy + max(ly - 1, 0) * abs(incy) + 3)
        ym = matrix(y.copy())
        base.gemv(A, matrix(x), ym, trans=trans, alpha=alpha, beta=beta, m=m, n=n, incx=incx, incy=incy,
                  offsetA=oi + oj * M, offsetx=ox, offsety=oy)
        out["c%d_x" % ci] = x; out["c%d_y" % ci] = y; out["c%d_out" % ci] = tolist(ym)
        cases.append([trans, m, n, oi + oj * M, incx, incy, ox, oy, alpha, beta])
    out["cases"] = np.array(json.dumps(cases))
    np.savez(os.path.join(HERE, "g13_gemv_subblocks.npz"), **out)

def g17_conelp_refinement():
    """conelp with options['refinement'] = 1, 2 (coneprog.py:502-507 default 0 for an LP; :599-631 res(), :1211-1235 the
    refinement wrap of f6): pure reference, dense-G LAPACK branch.  An ill-scaled variant (rows of G scaled over six decades)
    makes the refinement steps change the iterates visibly."""
    from kvxopt import matrix, solvers, spmatrix
    from kvxopt_amd import workloads
    solvers.options["show_progress"] = False
    out, meta = {}, {}
    L = workloads.lp_grid(6, 5)
    E = workloads.lp_grid_eq(6, 5, 4)
    rng = np.random.default_rng(17)
    sc = 10.0 ** rng.uniform(-3, 3, L["ml"])
    cases = [("grid6x5", L, None), ("grid6x5_scaled", L, sc), ("eq6x5p4", E, None)]
    for name, P, rs in cases:
        ml, n = P["ml"], P["n"]
        cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
        gx = P["Gx"] * (rs[P["Gi"]] if rs is not None else 1.0)
        hh = P["h"] * (rs if rs is not None else 1.0)
        G = matrix(spmatrix(gx.tolist(), P["Gi"].tolist(), cols.tolist(), (ml, n)))
        kw = {}
        if "Ap" in P:
            acols = np.repeat(np.arange(n), np.diff(P["Ap"]))
            kw = {"A": matrix(spmatrix(P["Ax"].tolist(), P["Ai"].tolist(), acols.tolist(), (P["p"], n))), "b": matrix(P["b"])}
        if rs is not None:
            out[name + "_rowscale"] = rs
        for ref in (0, 1, 2):
            solvers.options["refinement"] = ref
            sol = solvers.conelp(matrix(P["c"]), G, matrix(hh), kktsolver="chol2", **kw)
            key = "%s_r%d" % (name, ref)
            for k in "xsz":
                out[key + "_" + k] = tolist(sol[k])
            meta[key] = {k: sol[k] for k in ("status", "iterations", "gap", "relative gap", "primal objective", "dual objective",
                                             "primal infeasibility", "dual infeasibility")}
    del solvers.options["refinement"]
    np.savez_compressed(os.path.join(HERE, "g17_conelp_refinement.npz"), **out)
    json.dump({"via": "reference (dense LAPACK branch of misc.kkt_chol2), options['refinement'] = 0, 1, 2", "cases": meta},
              open(os.path.join(HERE, "g17_conelp_refinement.json"), "w"), indent=1, default=float)


def g18_nt_scaling_long():
    """The NT-scaling 'l' operations at the vector length of BASELINE configs[3] (ml = 200 000): inputs are regenerated from the
    seed by the test; of every output the fixture keeps each 997th entry, the sum and the 2-norm."""
    from kvxopt import matrix, misc, misc_solvers
    ml = 200000
    rng = np.random.default_rng(100 + ml)
    s = rng.uniform(0.1, 3.0, ml); z = rng.uniform(0.1, 3.0, ml)
    dims = {"l": ml, "q": [], "s": []}
    lm = matrix(0.0, (ml, 1))
    W = misc.compute_scaling(matrix(s), matrix(z), lm, dims)
    full = {"d": tolist(W["d"]), "di": tolist(W["di"]), "lmbda": tolist(lm)}
    X = rng.standard_normal((ml, 2))
    for tr in "NT":
        for inv in "NI":
            x = matrix(X.copy(order="F"))
            misc_solvers.scale(x, W, trans=tr, inverse=inv)
            full["scale_%s%s" % (tr, inv)] = np.array(x).reshape(-1, order="F")
    x1 = rng.standard_normal(ml); y1 = rng.uniform(0.5, 2.0, ml)
    for name, fn in (("scale2_N", lambda a: misc_solvers.scale2(lm, a, dims)),
                     ("scale2_I", lambda a: misc_solvers.scale2(lm, a, dims, inverse="I")),
                     ("sprod", lambda a: misc_solvers.sprod(a, matrix(y1), dims)),
                     ("sinv", lambda a: misc_solvers.sinv(a, matrix(y1), dims))):
        a = matrix(x1.copy()); fn(a); full[name] = tolist(a)
    a = matrix(0.0, (ml, 1)); misc.ssqr(a, matrix(x1), dims); full["ssqr"] = tolist(a)
    ds = rng.uniform(0.2, 2.0, ml); dz = rng.uniform(0.2, 2.0, ml)
    W2 = {"d": matrix(tolist(W["d"])), "di": matrix(tolist(W["di"])), "v": [], "beta": [], "r": [], "rti": []}
    lm2 = matrix(tolist(lm)); ms, mz = matrix(ds.copy()), matrix(dz.copy())
    misc.update_scaling(W2, lm2, ms, mz)
    full.update({"us_s": tolist(ms), "us_z": tolist(mz), "us_d": tolist(W2["d"]), "us_di": tolist(W2["di"]), "us_lmbda": tolist(lm2)})
    out = {"ml": np.array(ml), "seed": np.array(100 + ml), "stride": np.array(997),
           "sdot": np.array(misc_solvers.sdot(matrix(x1), matrix(y1), dims)), "max_step": np.array(misc_solvers.max_step(matrix(x1), dims))}
    for k, v in full.items():
        v = np.asarray(v, dtype=float).reshape(-1)
        out[k + "_sample"] = v[::997].copy(); out[k + "_sum"] = np.array(v.sum()); out[k + "_nrm2"] = np.array(np.linalg.norm(v))
    np.savez_compressed(os.path.join(HERE, "g18_nt_scaling_long.npz"), **out)


if __name__ == "__main__":
    stage()
    import kvxopt
    print("reference kvxopt", kvxopt.__version__)
    if len(sys.argv) > 1:                     # e.g. `make_goldens.py g12_nonlinear_block`: only the named generators
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_nt_scaling(kvxopt)
    g2_assembly()
    g3_kkt()
    g4_conelp()
    g5_coneqp()
    g6_conelp_std()
    g7_mps()
    g8_conelp_eq()
    g9_coneqp_eq()
    g10_conelp_starts()
    g11_coneqp_initvals()
    g12_nonlinear_block()
    g13_gemv_subblocks()
    g14_kkt_singular()
    g15_q_cone_scaling()
    g16_s_cone_scaling()
    g17_conelp_refinement()
    g18_nt_scaling_long()
    print("goldens written to", HERE)
