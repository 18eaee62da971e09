#!/usr/bin/env python3
"""bench.py -- headline benchmark of the KKT factor/solve hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY 8(d) "Config 2"): supernodal Cholesky factor +
solve of the 5-point Laplacian on a 1000 x 1000 grid (n = 1e6, 2 998 000 stored lower
entries, int64 CCS), b = default_rng(2).standard_normal(n), nrhs = 1.

A "step" = one numeric factorisation (values already resident in HBM) + one solve A x = b
(b resident in HBM), on a pattern analysed once before the timed region -- the per-IPM-step
usage of misc.kkt_chol2 (numeric refactor with the symbolic analysis reused, misc.py:1462).
metric = (sum_j c_j^2 + 4 nnz(L) nrhs) / time  [GF/s], c_j from the library's own symbolic
analysis for the permutation it uses.

One JSON line on stdout (rank 0).  N > 1 ranks: every rank factors its own system (the
systems are independent; no data-path collective) -> "scaling": "weak".  `--dist subtree`
shards ONE system over the ranks instead (kvxopt_amd/dist.py) -> "scaling": "strong";
`--workload lap3d --grid 200` is BASELINE configs[4].  The line carries `roofline` (dominant
kernel family, timed live with HIP events on the stream it runs on) and `cpu_baseline` (host
supernodal restatement on all cores, SciPy SuperLU and the simplicial oracle beside it) in
both modes, and `ipm` (IPM iterations/s with its own roofline and CPU baseline).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6     # MI355X FP64 matrix = vector peak; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, default=0, help="grid side (default: 1000 for the 5-pt Laplacian of config 2, 100 for --workload lap3d)")
    ap.add_argument("--workload", default="lap2d", choices=["lap2d", "lap3d"],
                    help="lap2d = BASELINE configs[1] (5-pt Laplacian, the headline); lap3d = 7-pt Laplacian on a cube (configs[4] is --grid 200)")
    ap.add_argument("--dist-ob", type=int, default=0, help="--dist subtree: column-block width of the block-cyclic fronts (0 = library default)")
    ap.add_argument("--dist-min-m", type=int, default=0, help="--dist subtree: smallest order of a block-cyclic front (0 = library default)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the host supernodal baseline (0 = all cores)")
    ap.add_argument("--no-splu", action="store_true", help="skip the SciPy SuperLU line of the CPU baseline (it takes ~10 s on config 2)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ipm", action="store_true", help="skip the secondary IPM iterations/s measurement")
    ap.add_argument("--roofline-family", default="auto")
    ap.add_argument("--chol-opts", default="", help="JSON dict of analysis options (nd_leaf, leaf_cols, leaf_rows, relax_*) for experiments")
    ap.add_argument("--quick", action="store_true", help="skip the per-family roofline loop and the CPU baseline (experiments)")
    ap.add_argument("--dist", default="replicas", choices=["replicas", "subtree"],
                    help="N > 1 ranks: 'replicas' = every rank factors its own system (weak scaling, no data-path collective; default); "
                         "'subtree' = ONE system sharded over the ranks (kvxopt_amd.dist.DistFactor: subtrees by proportional mapping, "
                         "block-cyclic top fronts with panel broadcasts; strong scaling)")
    return ap.parse_args()


def front_stats(F):
    """Algorithmic work per kernel family from the symbolic analysis (host side)."""
    sup, nrows, parent, level = F.supernodes()
    k = np.diff(sup)
    m = nrows
    u = m - k
    small = (m <= 128) & (k <= 64)          # wave + LDS kernel classes (symbolic.hpp front_class)
    # children update-matrix entries read by each front = sum over its children of u_c(u_c+1)/2
    child_tri = np.zeros(len(m))
    has_parent = parent >= 0
    np.add.at(child_tri, parent[has_parent], (u[has_parent] * (u[has_parent] + 1) / 2))
    child_u = np.zeros(len(m))
    np.add.at(child_u, parent[has_parent], u[has_parent])
    # LDS-front kernel: panel read + write (m*k doubles each), children's lower triangles read,
    # own update lower triangle written, child relative indices (int32) read
    bytes_small = float(np.sum((2 * m * k + child_tri + u * (u + 1) / 2)[small]) * 8 + np.sum(child_u[small]) * 4)
    # blocked big-front trailing update: sum over 64-column panel steps of nbk * t * (t + 1) flops
    flops_syrk = 0.0
    for kk, mm in zip(k[~small], m[~small]):
        jb = 0
        while jb < kk:
            nbk = min(64, kk - jb)              # KVX_NB
            t = mm - jb - nbk
            flops_syrk += float(nbk) * t * (t + 1)
            jb += 64
    lsize = float(np.sum(m * k))
    sum_m = float(np.sum(m))
    mid = ~small & (m <= int(os.environ.get("KVX_MID_M", "0")))   # big-class fronts factored by one workgroup each (opt-in experiment)
    S2 = lambda v: v * (v + 1.0) * (2.0 * v + 1.0) / 6.0
    flops_mid = float(np.sum((S2(m.astype(float)) - S2((m - k).astype(float)))[mid]))
    return {"bytes_small": bytes_small, "flops_syrk": flops_syrk, "flops_mid": flops_mid, "lsize": lsize, "sum_m": sum_m,
            "n_small": int(small.sum()), "n_big": int((~small).sum())}


def host_threads(args):
    """Threads of the host baselines: the cores this process may run on (affinity mask), at most 32 -- the OpenBLAS inside scipy is
    built for 64 threads and its buffer table overflows when more OpenMP threads than that call it at once."""
    if args.cpu_threads:
        return max(1, min(int(args.cpu_threads), 32))
    try:
        c = len(os.sched_getaffinity(0))
    except Exception:
        c = os.cpu_count() or 1
    return max(1, min(c, 16))                     # 16 = the CPU share that goes with one GPU of this node


def cpu_baselines(args, F, n, colptr, rowind, values, b_host, nrhs, work, x_gpu):
    """The reference's CPU path is SuiteSparse CHOLMOD (third-party, absent from this image: probed below).  Timed instead, on
    the same matrix, permutation and flop count: (1) the host supernodal multifrontal restatement on all cores (OpenMP
    subtrees + OpenBLAS BLAS-3 in the fronts, oracle/kvx_supernodal.c) -- the strongest CPU number and the headline baseline;
    (2) SciPy SuperLU on P A P' with the natural column order, 1 thread (the calibration of BASELINE.md section 2);
    (3) the simplicial up-looking oracle, 1 core (the parity checker)."""
    import ctypes.util
    B = np.asfortranarray(b_host.reshape(n, nrhs, order="F").copy())
    cores = host_threads(args)
    out = {"cholmod_found": bool(ctypes.util.find_library("cholmod")), "os_cpu_count": os.cpu_count(), "affinity_cores": (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None),
           "env_threads": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS") if os.environ.get(k)}}
    others = {}
    try:
        from oracle.kvx_oracle import OracleSupernodal
        S = OracleSupernodal.from_factor(n, colptr, rowind, F, threads=cores)
        best = None
        for _ in range(3):                              # first pass pays page faults and thread start-up: best of three
            xb = B.copy(order="F")
            t0 = time.perf_counter(); S.factorize(values); t1 = time.perf_counter(); S.solve(xb); t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        diff = float(np.abs(xb.reshape(-1, order="F") - x_gpu).max() / np.abs(xb).max())
        out.update({"value": work / best[0] / 1e9, "unit": "GF/s", "cores": cores, "kind": "port",
                    "library": "host supernodal multifrontal restatement (oracle/kvx_supernodal.c): OpenMP over elimination-tree subtrees, "
                               "OpenBLAS dpotrf/dtrsm/dsyrk inside the fronts (the OpenBLAS that ships in scipy)",
                    "sample": "same system, permutation and supernodes as the GPU run: numeric factorisation %.3f s + solve %.3f s, best of 3"
                              % (best[1], best[2]),
                    "max_rel_diff_vs_gpu": diff})
        del S
    except Exception as e:
        out["supernodal_error"] = repr(e)
    if not args.no_splu and n <= 1200000:
        try:
            import scipy.sparse as sp
            from scipy.sparse.linalg import splu
            perm = F.perm()
            A = sp.csc_matrix((values, rowind, colptr), shape=(n, n))
            A = (A + sp.tril(A, -1).T).tocsc()
            Ap = A[perm][:, perm].tocsc()
            t0 = time.perf_counter()
            lu = splu(Ap, permc_spec="NATURAL", diag_pivot_thresh=0.0, options={"SymmetricMode": True})
            t1 = time.perf_counter()
            xs = lu.solve(B[perm])
            t2 = time.perf_counter()
            xs_full = np.empty_like(xs); xs_full[perm] = xs
            others["scipy_splu"] = {"value": work / (t2 - t0) / 1e9, "unit": "GF/s", "cores": 1, "kind": "port",
                                    "library": "scipy.sparse.linalg.splu (SuperLU), permc_spec=NATURAL on P A P', diag_pivot_thresh=0, SymmetricMode",
                                    "factor_s": t1 - t0, "solve_s": t2 - t1, "nnz_L": int(lu.L.nnz),
                                    "max_rel_diff_vs_gpu": float(np.abs(xs_full.reshape(-1, order="F") - x_gpu).max() / np.abs(xs_full).max())}
            del lu, Ap, A
        except Exception as e:
            others["scipy_splu"] = {"error": repr(e)}
    try:
        from oracle.kvx_oracle import OracleChol
        O = OracleChol(n, colptr, rowind, "L", F.perm())
        t0 = time.perf_counter()
        O.factorize(values)
        xb = B.copy(order="F")
        O.solve(xb)
        tc = time.perf_counter() - t0
        others["simplicial_oracle"] = {"value": work / tc / 1e9, "unit": "GF/s", "cores": 1, "kind": "port",
                                       "library": "oracle/kvx_oracle.c (up-looking simplicial, the parity checker)", "seconds": tc,
                                       "max_rel_diff_vs_gpu": float(np.abs(xb.reshape(-1, order="F") - x_gpu).max() / np.abs(xb).max())}
    except Exception as e:
        others["simplicial_oracle"] = {"error": repr(e)}
    out["others"] = others
    if "value" not in out:                               # the supernodal library could not be built: fall back to the best other line
        ok = [v for v in others.values() if "value" in v]
        if ok:
            bestv = max(ok, key=lambda v: v["value"])
            out.update({k: bestv[k] for k in ("value", "unit", "cores", "kind", "library")})
            out["sample"] = "same system and permutation, 1 numeric factorisation + 1 solve"
    return out


def ipm_leg(args, torch):
    """IPM iterations/s of the device-resident conelp on BASELINE configs[3] (inequality form), with the roofline of the
    normal-equations assembly kernel (k_atda, HBM-bound: 12 nnz(G) + 8 ml + 8 nnz(S) bytes per launch) timed
    live with events on the stream it runs on (the null stream), and a CPU run of the same loop beside it."""
    from kvxopt_amd import _lib
    from kvxopt_amd import lp as kvx_lp
    from kvxopt_amd import workloads
    from kvxopt_amd.base import spmatrix
    Pl = workloads.lp_grid(250, 200)
    ml, nl = Pl["ml"], Pl["n"]
    Gl = spmatrix.from_ccs(ml, nl, Pl["Gp"], Pl["Gi"], Pl["Gx"])
    kvx_lp.conelp(Pl["c"], Gl, Pl["h"], options={"maxiters": 2})       # code objects, pools
    kvx_lp.clear_cache()
    t00 = time.perf_counter()
    s0 = kvx_lp.conelp(Pl["c"], Gl, Pl["h"])                           # a NEW constraint structure: analysis, set-up, graph capture
    t0c = time.perf_counter() - t00
    tl0 = time.perf_counter()
    sl = kvx_lp.conelp(Pl["c"], Gl, Pl["h"])                           # the same structure again: KKT objects kept (lp._kkt_for)
    tl = time.perf_counter() - tl0
    ipm = {"metric": "IPM iterations/s", "value": sl["iterations"] / sl["loop seconds"], "unit": "iterations/s",
           "workload": "conelp, grid LP 250x200: ml=200000 inequalities, n=50000 (BASELINE configs[3], inequality form)",
           "measured": "interior-point loop (coneprog.py:859-1436 equivalent) of a call on a constraint structure seen before -- the "
                       "steady state of a sequence of LPs on fixed patterns, as the headline step reuses its symbolic analysis; "
                       "first_call = a new structure (analysis, device set-up and launch-graph capture inside the call)",
           "iterations": sl["iterations"], "status": sl["status"], "loop_s": sl["loop seconds"], "whole_call_s": tl,
           "value_whole_call": sl["iterations"] / tl,
           "first_call": {"value": s0["iterations"] / s0["loop seconds"], "loop_s": s0["loop seconds"], "whole_call_s": t0c,
                          "value_whole_call": s0["iterations"] / t0c, "iterations": s0["iterations"]}}
    # roofline of the assembly kernel: S = G' diag(w) G on the fixed pattern, launched back to back on the null stream
    L = _lib.lib()
    h = ctypes.c_void_p()
    _lib.raise_for(L.kvx_atda_plan(ml, nl, _lib.pi(Pl["Gp"]), _lib.pi(Pl["Gi"]), None, None, ctypes.byref(h)))
    snz = ctypes.c_int64()
    _lib.raise_for(L.kvx_atda_pattern(h, ctypes.byref(snz), None, None))
    gx = torch.from_numpy(Pl["Gx"]).cuda(); w = torch.rand(ml, dtype=torch.float64, device="cuda") + 0.5
    sx = torch.zeros(max(snz.value, 1), dtype=torch.float64, device="cuda")
    reps = 50
    for _ in range(5):
        _lib.raise_for(L.kvx_atda_assemble_dev(h, gx.data_ptr(), w.data_ptr(), None, sx.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        _lib.raise_for(L.kvx_atda_assemble_dev(h, gx.data_ptr(), w.data_ptr(), None, sx.data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 12.0 * len(Pl["Gx"]) + 8.0 * ml + 8.0 * snz.value
    ipm["roofline"] = {"kernel": "k_atda (S = G' diag(w) G on the fixed pattern)", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "launches_timed": reps,
                       "note": "back-to-back launches: time per launch includes the launch gap; the loop itself is bound by the factor/solve latency chains"}
    L.kvx_atda_free(h)
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r02_ipm_pmc.json")))
        ipm["roofline"]["traffic"] = pm.get("k_atda_bytes_per_launch")
    except Exception:
        pass
    # CPU run of the same loop (oracle/lp_oracle.py: coneprog.py:859-1436 restated over the host supernodal Cholesky)
    if not args.no_cpu_baseline:
        try:
            import scipy.sparse as sp
            from kvxopt_amd.chol import Factor
            from oracle import lp_oracle
            G = sp.csc_matrix((Pl["Gx"], Pl["Gi"], Pl["Gp"]), shape=(ml, nl))
            Sp_ = sp.tril((abs(G.T) @ abs(G)).tocsc()).tocsc(); Sp_.sort_indices()
            Sp, Si = Sp_.indptr.astype(np.int64), Sp_.indices.astype(np.int64)
            Fs = Factor(nl, Sp, Si)
            sup, nrows, parent, level = Fs.supernodes()
            rp, ri = Fs.front_rows()
            cores = host_threads(args)
            sc = lp_oracle.conelp_l(Pl["c"], ml, nl, Pl["Gp"], Pl["Gi"], Pl["Gx"], Pl["h"], structure=(Fs.perm(), sup, rp, ri, parent, Sp, Si),
                                    threads=cores)
            ipm["cpu_baseline"] = {"value": sc["iterations"] / sc["loop seconds"], "unit": "iterations/s", "cores": cores, "kind": "port",
                                   "library": "oracle/lp_oracle.py (coneprog.py:859-1436 restated, numpy) over the host supernodal Cholesky "
                                              "(oracle/kvx_supernodal.c) and the 1-thread C assembly of S (oracle/kvx_oracle.c)",
                                   "sample": "the whole run: %d iterations in %.2f s (assembly %.2f s, factor %.2f s, solves %.2f s)"
                                             % (sc["iterations"], sc["loop seconds"], sc["assemble_s"], sc["factor_s"], sc["solve_s"]),
                                   "iterations": sc["iterations"], "status": sc["status"],
                                   "max_abs_diff_x_vs_gpu": float(np.abs(np.asarray(sc["x"]) - np.asarray(sl["x"]).reshape(-1)).max())}
            ipm["vs_cpu_baseline"] = ipm["value"] / ipm["cpu_baseline"]["value"]
        except Exception as e:
            ipm["cpu_baseline"] = {"error": repr(e)}
    return ipm


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # one GPU per rank; KVX_DIST_BACKEND=gloo rehearses the N > 1 paths on a box with fewer GPUs than ranks
    backend = os.environ.get("KVX_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor

    g = args.grid or (1000 if args.workload == "lap2d" else 100)
    if args.workload == "lap2d":
        n, colptr, rowind, values = workloads.laplacian_2d(g)
        wname = "5-pt Laplacian %dx%d" % (g, g)
    else:
        n, colptr, rowind, values = workloads.laplacian_3d(g)
        wname = "7-pt Laplacian %dx%dx%d" % (g, g, g)
    chol_opts = json.loads(args.chol_opts) if args.chol_opts else None
    DF = None
    t0 = time.time()
    if args.dist == "subtree":
        from kvxopt_amd.dist import DistFactor
        DF = DistFactor(n, colptr, rowind, "L", None, chol_opts, device=torch.device("cuda", local_rank), ob=args.dist_ob, min_m=args.dist_min_m)
        F = DF.F
    else:
        F = Factor(n, colptr, rowind, "L", None, chol_opts)
    t_analyze = time.time() - t0
    info = F.info()
    nrhs = args.nrhs
    work = info["flops"] + 4.0 * info["lnz"] * nrhs          # SURVEY 8(d) flop measure per step

    dev = torch.device("cuda", local_rank)
    vals_d = torch.from_numpy(values).to(dev)
    b_host = np.random.default_rng(2).standard_normal((n, nrhs)).reshape(n * nrhs, order="F") if nrhs > 1 \
        else np.random.default_rng(2).standard_normal(n)
    b_d = torch.from_numpy(np.ascontiguousarray(b_host)).to(dev)
    x_d = torch.empty_like(b_d)
    torch.cuda.synchronize()

    def step():
        if DF is not None:                                   # one system over all ranks
            DF.factorize(vals_d)
            x_d.copy_(b_d)
            DF.solve(x_d, nrhs)
            return
        F.factorize_dev(vals_d.data_ptr(), sync=False)
        x_d.copy_(b_d)
        torch.cuda.current_stream().synchronize()          # x_d ready before the factor's own stream reads it
        F.solve_dev(x_d.data_ptr(), 0, nrhs, n)              # synchronises the factor's stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # --- timed region ------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_factor, ms_solve = F.timing()
    steps_run = args.warmup + args.steps
    if DF is not None:
        steps_run_marker = (DF.collectives, DF.bytes_moved)   # (counted up to here: warm-up + timed steps)

    # residual of the last solve (parity bar: <= 1e-10 relative)
    x = x_d.cpu().numpy()
    r = workloads.sym_matvec(n, colptr, rowind, values, x.reshape(n, nrhs, order="F")) - b_host.reshape(n, nrhs, order="F")
    relres = float(np.linalg.norm(r) / np.linalg.norm(b_host))

    # --- roofline leg: dominant kernel family timed live with HIP events on the factor's stream
    st = front_stats(F)
    fam_times = {}
    for fam in (Factor.FAMILIES if not args.quick else ()):
        samples = []
        for _ in range(3):                         # median of three single-step readings: one hiccup must not pick the family
            F.prof_select(fam)
            step()
            samples.append(F.prof_read())
        samples.sort(key=lambda t: t[0])
        fam_times[fam] = (samples[1][0], samples[1][1])
    F.prof_select(None)
    if args.quick:
        if rank == 0:
            print(json.dumps({"value": work * args.steps * world / dt / 1e9, "ms_per_step": dt / args.steps * 1e3, "ms_factor": ms_factor,
                              "ms_solve": ms_solve, "rel_residual": relres, "nsuper": int(info["nsuper"]), "nlevels": int(info["nlevels"]),
                              "lsize": int(info["lsize"]), "opts": args.chol_opts}))
        return
    dom = max(fam_times, key=lambda f: fam_times[f][0]) if args.roofline_family == "auto" else args.roofline_family
    dom_ms, dom_launches = fam_times[dom]
    if DF is not None and world > 1:
        # rank 0 times its own launches: scale the family's work by the share of the flops rank 0 executes (host-side map)
        from kvxopt_amd.dist import partition
        mp = partition(F, world, DF.ob, DF.min_m)
        share0 = float(mp["rank_flops"][0] / mp["flops"])
        st = {k: (v * share0 if isinstance(v, float) else v) for k, v in st.items()}
    if dom in ("syrk_trailing", "front_mid"):
        achieved = st["flops_syrk" if dom == "syrk_trailing" else "flops_mid"] / (dom_ms * 1e-3) / 1e12
        roofline = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": achieved / FP64_MFMA_PEAK_TF, "traffic": None}
    else:
        if dom == "front_small":
            alg_bytes = st["bytes_small"]
        elif dom in ("fwd_level", "bwd_level"):
            alg_bytes = (8.0 * st["lsize"] + 4.0 * st["sum_m"] + 16.0 * n) * nrhs
        elif dom == "scatter_a":
            alg_bytes = 24.0 * len(values)
        else:
            alg_bytes = 8.0 * st["lsize"]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None}
    # HBM traffic of the dominant family from the committed PMC passes (profiles/, separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); null when no profile matches.
    try:
        pmc_file = "r02_pmc_fetch_write_per_kernel.json"
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        if args.workload == "lap2d" and g == 1000 and nrhs == 1 and DF is None and dom in pmc.get("family_bytes_per_step", {}):
            roofline["traffic"] = pmc["family_bytes_per_step"][dom] / max(dom_launches, 1)
            roofline["traffic_unit"] = "bytes per launch (FETCH_SIZE+WRITE_SIZE, raw; profiles/%s)" % pmc_file
            roofline["algorithmic_bytes_per_launch"] = (alg_bytes / max(dom_launches, 1)) if roofline["bound"] == "hbm" else None
    except Exception:
        pass
    roofline["ms_per_step"] = dom_ms
    roofline["launches_per_step"] = dom_launches
    roofline["family_ms_per_step"] = {f: round(v[0], 4) for f, v in fam_times.items()}

    # --- CPU baselines on the same system and permutation (rank 0, host cores of this box) -------
    cpu = None
    if rank == 0 and (world == 1 or DF is not None) and not args.no_cpu_baseline:   # (replicas: rank 0 at N = 1 only; one sharded system: beside every N)
        cpu = cpu_baselines(args, F, n, colptr, rowind, values, b_host, nrhs, work, x)

    # --- the other half of BASELINE.json's metric: IPM iterations/s of the device-resident conelp on configs[3]
    # (inequality form), rank 0 only, a few hundred ms; never part of `value`
    ipm = None
    if rank == 0 and world == 1 and not args.no_ipm:
        try:
            ipm = ipm_leg(args, torch)
        except Exception as e:                      # the headline line must not depend on this leg
            ipm = {"error": repr(e)}

    if rank == 0:
        total = work * args.steps * (1 if DF is not None else world)
        out = {
            "metric": "sparse Cholesky factor+solve GF/s", "value": total / dt / 1e9, "unit": "GF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if DF is not None else "weak",
            "vs_baseline": None,                    # BASELINE.md holds no published number for this metric
            "vs_cpu_baseline": (total / dt / 1e9 / cpu["value"]) if (cpu and cpu.get("value")) else None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, n=%d, lower CCS int64, nrhs=%d, factor+solve per step" % (wname, n, nrhs),
                       "nnz_lower": int(len(values)), "lnz": int(info["lnz"]), "flops_sum_cj2": info["flops"],
                       "nsuper": int(info["nsuper"]), "nlevels": int(info["nlevels"]), "max_front": int(info["max_front"]),
                       "analyze_s": round(t_analyze, 3),
                       "parallelism": ("replicas x%d" % world) if DF is None else
                                      ("one system sharded x%d: proportional mapping, %d shared fronts on rank 0 (%d block-cyclic, %d-column blocks), "
                                       "%d collectives / %.1f MB per step on rank 0" % (world, DF.nshared, DF.ncyclic, DF.ob,
                                                                                        steps_run_marker[0] // max(steps_run, 1), steps_run_marker[1] / max(steps_run, 1) / 1e6))},
            "ms_factor": ms_factor, "ms_solve": ms_solve, "rel_residual": relres,
            "roofline": roofline, "cpu_baseline": cpu, "ipm": ipm,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()                  # rank 0 has extra legs (CPU baseline, IPM): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
