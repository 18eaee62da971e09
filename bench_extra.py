#!/usr/bin/env python3
"""bench_extra.py -- the other measured configurations of the hot path (one JSON line each, 1 GPU).

  chol5   : bench.py's workload again with nrhs = 3 (SURVEY 8(d) config 2 "also nrhs=3")
  chol21  : the "~20 nnz/row" north-star variant, 21-point stencil on the 1000 x 1000 grid (workloads.stencil21_2d)
  lap3d   : 7-point Laplacian on a 100^3 grid (n = 1e6; one eighth of config 5's unknowns): the flop-bound regime
            (nnz(L) = 5.4e8, 2.7e12 flops, top front of order 14 082)
  lap3d200: BASELINE.json configs[4] (7-point Laplacian 200^3, n = 8e6) on ONE GPU -- not part of the default run
  lp4a    : BASELINE.json configs[3] literally: standard form, 50 000 equality rows, 200 000
            variables, G = -I; the equality branch of misc.kkt_chol2 (K = A S^-1 A' refactored every iteration)
  lp4b    : BASELINE.json configs[3] in inequality form: the device-resident
            interior-point loop kvxopt_amd.lp.conelp on the 250 x 200 grid LP (ml = 200 000, n = 50 000);
            metric = IPM iterations per second (wall time of the whole conelp call, analysis included),
            plus the per-iteration cost of one KKT factor (assembly + numeric Cholesky) and one KKT solve.

  chol5x64, chol5x256 : config 2 with 64 / 256 right-hand sides (solves on rhs-major blocks, csrc/kernels_wide.hip)
  lp4c    : lp4b plus 200 random equality rows: general G and A together (dense K = A S^-1 A', lp.KKTGenEqDev)
  klu3    : BASELINE.json configs[2]: klu.linsolve on the ACTIVSg2000 power-flow Jacobian (4000 x 4000, 29 336 entries,
            tests/golden/ACTIVSg2000.npz), nrhs = 3: symbolic, first numeric, steady-state refactorisation and solve times,
            next to SciPy's SuperLU on the host (the CPU oracle is test infrastructure and is not used here).
  lu2d    : unsymmetric counterpart of config 2 for the LU path: convection-diffusion 5-point operator on a 600 x 600 grid
            (n = 360 000, workloads.convdiff_2d): refactorisation + solve on the GPU next to SciPy's SuperLU on the host.

The headline line the driver reads is bench.py's; this script documents the rest (results in DESIGN.md section 5).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def chol_case(name, n, colptr, rowind, values, nrhs, steps, warmup):
    import torch
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    t0 = time.time()
    F = Factor(n, colptr, rowind, "L", None, None)
    t_an = time.time() - t0
    info = F.info()
    dev = torch.device("cuda", 0)
    vals_d = torch.from_numpy(values).to(dev)
    B = np.asfortranarray(np.random.default_rng(2).standard_normal((n, nrhs)))
    b_d = torch.from_numpy(B.reshape(-1, order="F").copy()).to(dev)
    x_d = torch.empty_like(b_d)

    def step():
        F.factorize_dev(vals_d.data_ptr(), sync=False)
        x_d.copy_(b_d)
        torch.cuda.current_stream().synchronize()
        F.solve_dev(x_d.data_ptr(), 0, nrhs, n)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    msf, mss = F.timing()
    X = x_d.cpu().numpy().reshape(n, nrhs, order="F")
    r = workloads.sym_matvec(n, colptr, rowind, values, X) - B
    work = info["flops"] + 4.0 * info["lnz"] * nrhs
    return {"case": name, "metric": "sparse Cholesky factor+solve GF/s", "value": work / dt / 1e9, "unit": "GF/s",
            "ms_per_step": dt * 1e3, "ms_factor": msf, "ms_solve": mss, "nrhs": nrhs, "n": n, "nnz_lower": int(len(values)),
            "lnz": int(info["lnz"]), "flops_sum_cj2": info["flops"], "nsuper": int(info["nsuper"]), "nlevels": int(info["nlevels"]),
            "max_front": int(info["max_front"]), "analyze_s": round(t_an, 3),
            "rel_residual": float(np.linalg.norm(r) / np.linalg.norm(B))}


def lp_case(gx, gy, repeat=1, kkt_timing=True):
    from kvxopt_amd import lp, workloads
    from kvxopt_amd.base import spmatrix
    P = workloads.lp_grid(gx, gy)
    ml, n = P["ml"], P["n"]
    cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
    G = spmatrix(P["Gx"], P["Gi"], cols, (ml, n))
    lp.conelp(P["c"], G, P["h"], options={"maxiters": 2})            # warm-up: HIP module load, first-touch allocations
    for _ in range(max(1, repeat)):                                 # (--repeat: several whole runs under a profiler)
        t0 = time.perf_counter()
        sol = lp.conelp(P["c"], G, P["h"])
        dt = time.perf_counter() - t0
    if not kkt_timing:
        return {"case": "lp4b", "metric": "IPM iterations/s", "value": sol["iterations"] / dt, "unit": "iterations/s",
                "iterations": sol["iterations"], "status": sol["status"], "wall_s": dt, "loop_s": sol["loop seconds"], "runs": repeat}
    # per-iteration KKT costs on the same pattern
    kkt = lp.KKTChol2Dev(ml, n, P["Gp"], P["Gi"], P["Gx"])
    di = lp.DVec(ml, np.random.default_rng(4).uniform(0.2, 5.0, ml))
    xv, zv = lp.DVec(n, np.ones(n)), lp.DVec(ml, np.ones(ml))
    for _ in range(3):
        kkt.factor(di); kkt.solve(xv, zv)
    from kvxopt_amd import _lib
    sync = _lib.lib().kvx_dev_sync
    sync()
    t0 = time.perf_counter()
    for _ in range(20):
        kkt.factor(di)
    sync()
    t_f = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        kkt.solve(xv, zv)
    sync()
    t_s = (time.perf_counter() - t0) / 20
    return {"case": "lp4b", "metric": "IPM iterations/s", "value": sol["iterations"] / dt, "unit": "iterations/s",
            "iterations": sol["iterations"], "status": sol["status"], "wall_s": dt, "loop_s": sol["loop seconds"],
            "iterations_per_s_loop_only": sol["iterations"] / sol["loop seconds"] if sol["loop seconds"] > 0 else None,
            "ml": ml, "n": n,
            "gap": sol["gap"], "primal_infeasibility": sol["primal infeasibility"], "dual_infeasibility": sol["dual infeasibility"],
            "factorizations": sol["factorizations"], "ms_kkt_factor": t_f * 1e3, "ms_kkt_solve": t_s * 1e3}


def lp_std_case(gx, gy):
    from kvxopt_amd import lp, workloads
    from kvxopt_amd.base import spmatrix
    L = workloads.lp_grid_std(gx, gy)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    lp.conelp(L["c"], G, L["h"], A=A, b=L["b"], options={"maxiters": 2})
    t0 = time.perf_counter()
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    dt = time.perf_counter() - t0
    return {"case": "lp4a", "metric": "IPM iterations/s", "value": sol["iterations"] / dt, "unit": "iterations/s",
            "iterations": sol["iterations"], "status": sol["status"], "wall_s": dt, "loop_s": sol["loop seconds"],
            "iterations_per_s_loop_only": sol["iterations"] / sol["loop seconds"] if sol["loop seconds"] > 0 else None,
            "equality_rows": L["p"], "variables": L["n"], "gap": sol["gap"],
            "primal_infeasibility": sol["primal infeasibility"], "dual_infeasibility": sol["dual infeasibility"],
            "factorizations": sol["factorizations"]}


def lp_eq_case(gx, gy, p):
    """Config-4b grid LP plus p equality rows: general G AND A (lp.KKTGenEqDev: dense K = A S^-1 A')."""
    from kvxopt_amd import lp, workloads
    from kvxopt_amd.base import spmatrix
    L = workloads.lp_grid_eq(gx, gy, p)
    G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
    lp.conelp(L["c"], G, L["h"], A=A, b=L["b"], options={"maxiters": 2})
    t0 = time.perf_counter()
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
    dt = time.perf_counter() - t0
    return {"case": "lp4c grid %dx%d + %d equalities (general G)" % (gx, gy, p), "metric": "IPM iterations/s", "value": sol["iterations"] / dt,
            "unit": "iterations/s", "iterations": sol["iterations"], "status": sol["status"], "wall_s": dt, "loop_s": sol.get("loop seconds"),
            "iterations_per_s_loop_only": sol["iterations"] / sol["loop seconds"] if sol.get("loop seconds") else None,
            "ml": L["ml"], "n": L["n"], "p": p, "gap": sol["gap"], "primal_infeasibility": sol["primal infeasibility"],
            "dual_infeasibility": sol["dual infeasibility"]}


def klu_case(steps, warmup):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from kvxopt_amd import klu, _lib
    from kvxopt_amd.base import spmatrix
    z = np.load(os.path.join(ROOT, "tests", "golden", "ACTIVSg2000.npz"))
    n = int(z["n"])
    A = spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])
    As = sp.csc_matrix((z["values"], z["rowind"], z["colptr"]), shape=(n, n))
    B = np.asfortranarray(np.random.default_rng(3).standard_normal((n, 3)))
    t0 = time.perf_counter(); Fs = klu.symbolic(A); t_sym = time.perf_counter() - t0
    t0 = time.perf_counter(); Fn = klu.numeric(A, Fs); t_first = time.perf_counter() - t0
    vals_d = _lib.DeviceBuffer.from_array(A.values)
    b_d = _lib.DeviceBuffer.from_array(B.reshape(-1, order="F"))
    def timed(fn):
        for _ in range(warmup): fn()
        t0 = time.perf_counter()
        for _ in range(steps): fn()
        return (time.perf_counter() - t0) / steps * 1e3
    ms_refactor = timed(lambda: Fn.num.refactor_dev(vals_d.ptr, A.values.size))
    ms_solve = timed(lambda: Fn.num.solve_dev(b_d.ptr, "N", 3))
    ms_tsolve = timed(lambda: Fn.num.solve_dev(b_d.ptr, "T", 3))
    def lins():
        X = B.copy(order="F"); klu.linsolve(A, X); return X
    klu._LINSOLVE_CACHE.clear()
    t0 = time.perf_counter(); lins(); ms_linsolve_first = (time.perf_counter() - t0) * 1e3     # analysis + factorisation + solve
    ms_linsolve = timed(lins)                                # repeated calls on a known pattern: refactorisation + solve
    X = lins()
    resid = float(np.abs(As @ X - B).max())
    t0 = time.perf_counter(); lu = spla.splu(As); xs = lu.solve(B); t_splu = time.perf_counter() - t0
    e = Fn.num.extract()
    return {"case": "klu3 ACTIVSg2000", "metric": "klu.linsolve wall ms (host buffers; repeated call on a known pattern: cached analysis, "
                                          "refactorisation + solve)", "value": ms_linsolve, "ms_linsolve_first_call": ms_linsolve_first,
            "unit": "ms", "n": n, "nnz": int(A.values.size), "nrhs": 3, "ms_symbolic_host": t_sym * 1e3, "ms_first_numeric": t_first * 1e3,
            "ms_refactor_dev": ms_refactor, "ms_solve_dev": ms_solve, "ms_tsolve_dev": ms_tsolve, "residual_inf": resid,
            "lnz": int(e["L"][1].size), "unz": int(e["U"][1].size), **{"lu_" + k: v for k, v in Fn.num.info().items()},
            "cpu_scipy_superlu_ms": t_splu * 1e3, "x_vs_superlu": float(np.abs(X - xs).max())}


def lu2d_case(g, steps):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from kvxopt_amd import klu, _lib, workloads
    from kvxopt_amd.base import spmatrix
    n, cp, ri, v = workloads.convdiff_2d(g)
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    As = sp.csc_matrix((v, ri, cp), shape=(n, n))
    t0 = time.perf_counter(); Fs = klu.symbolic(A); t_sym = time.perf_counter() - t0
    t0 = time.perf_counter(); Fn = klu.numeric(A, Fs); t_first = time.perf_counter() - t0
    vals_d = _lib.DeviceBuffer.from_array(v)
    b = np.random.default_rng(1).standard_normal(n)
    b_d = _lib.DeviceBuffer.from_array(b)
    for _ in range(3): Fn.num.refactor_dev(vals_d.ptr, v.size)     # warm-up: the second call on the same buffers captures the launch graph
    t0 = time.perf_counter()
    for _ in range(steps): Fn.num.refactor_dev(vals_d.ptr, v.size)
    ms_ref = (time.perf_counter() - t0) / steps * 1e3
    for _ in range(3): b_d.upload(b); Fn.num.solve_dev(b_d.ptr, "N", 1)
    t0 = time.perf_counter()
    for _ in range(steps): b_d.upload(b); Fn.num.solve_dev(b_d.ptr, "N", 1)
    ms_solve = (time.perf_counter() - t0) / steps * 1e3
    x = b_d.download(np.float64, n)
    t0 = time.perf_counter(); lu = spla.splu(As); xs = lu.solve(b); t_splu = time.perf_counter() - t0
    e = Fn.num.extract()
    return {"case": "lu2d convection-diffusion %dx%d" % (g, g), "metric": "LU refactor+solve ms", "value": ms_ref + ms_solve, "unit": "ms",
            "n": n, "nnz": int(v.size), "ms_symbolic_host": t_sym * 1e3, "ms_first_numeric": t_first * 1e3, "ms_refactor_dev": ms_ref,
            "ms_solve_dev_incl_upload": ms_solve, "lnz": int(e["L"][1].size), "unz": int(e["U"][1].size),
            "rel_residual": float(np.linalg.norm(As @ x - b) / np.linalg.norm(b)), "merges": Fs.sym.info()["merges"],
            **{"lu_" + k: val for k, val in Fn.num.info().items()}, "cpu_scipy_superlu_ms": t_splu * 1e3,
            "x_vs_superlu": float(np.abs(x - xs).max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="chol5,chol21,lap3d,lp4a,lp4b")
    ap.add_argument("--grid", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeat", type=int, default=0, help="lp4b: this many whole conelp runs and nothing else (profiling runs)")
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_extra.py needs a HIP device (no CPU fallback)")
    from kvxopt_amd import workloads
    for case in args.cases.split(","):
        if case == "chol5":
            out = chol_case("chol5 nrhs=3", *workloads.laplacian_2d(args.grid), 3, args.steps, args.warmup)
        elif case == "chol5x64":                  # many right-hand sides: the rhs-major MFMA path (kernels_wide.hip)
            out = chol_case("chol5 nrhs=64", *workloads.laplacian_2d(args.grid), 64, max(3, args.steps // 2), 2)
        elif case == "chol5x256":
            out = chol_case("chol5 nrhs=256", *workloads.laplacian_2d(args.grid), 256, max(3, args.steps // 2), 2)
        elif case == "chol21":
            out = chol_case("chol21 (21-point stencil)", *workloads.stencil21_2d(args.grid), 1, args.steps, args.warmup)
        elif case == "lap3d":
            out = chol_case("lap3d 100^3", *workloads.laplacian_3d(100), 1, max(2, args.steps // 3), 1)
        elif case == "lap3d200":                  # BASELINE configs[4] on ONE GPU: the whole factor (nnz(L) ~ 9e9) sits in HBM
            out = chol_case("lap3d 200^3 (config 5, 1 GPU)", *workloads.laplacian_3d(200), 1, 2, 1)
        elif case == "lp4a":
            out = lp_std_case(250, 200)
        elif case == "lp4b":
            out = lp_case(250, 200, args.repeat, kkt_timing=args.repeat == 0)
        elif case == "lp4c":
            out = lp_eq_case(250, 200, 200)
        elif case == "lu2d":
            out = lu2d_case(600, max(3, args.steps // 2))
        elif case == "klu3":
            out = klu_case(args.steps * 5, args.warmup)
        else:
            raise SystemExit("unknown case " + case)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
