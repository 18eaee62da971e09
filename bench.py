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
systems are independent; no data-path collective) -> "scaling": "weak".
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1000, help="grid side of the 5-pt Laplacian (config 2: 1000)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ipm", action="store_true", help="skip the secondary IPM iterations/s measurement")
    ap.add_argument("--roofline-family", default="auto")
    ap.add_argument("--chol-opts", default="", help="JSON dict of analysis options (nd_leaf, leaf_cols, leaf_rows, relax_*) for experiments")
    ap.add_argument("--quick", action="store_true", help="skip the per-family roofline loop and the CPU baseline (experiments)")
    ap.add_argument("--dist", default="replicas", choices=["replicas", "subtree"],
                    help="N > 1 ranks: 'replicas' = every rank factors its own system (weak scaling, no data-path collective; default); "
                         "'subtree' = ONE system sharded by elimination-tree subtrees over the ranks (kvxopt_amd.dist.DistFactor: "
                         "all-reduce of the subtree-root update matrices per factorisation, of update vectors and x per solve; strong scaling)")
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
    return {"bytes_small": bytes_small, "flops_syrk": flops_syrk, "lsize": lsize, "sum_m": sum_m,
            "n_small": int(small.sum()), "n_big": int((~small).sum())}


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

    g = args.grid
    n, colptr, rowind, values = workloads.laplacian_2d(g)
    t0 = time.time()
    F = Factor(n, colptr, rowind, "L", None, json.loads(args.chol_opts) if args.chol_opts else None)
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

    DF = None
    if args.dist == "subtree" and world > 1:
        from kvxopt_amd.dist import DistFactor
        DF = DistFactor(n, colptr, rowind, "L", None, json.loads(args.chol_opts) if args.chol_opts else None, device=dev)

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
    if DF is not None:
        if rank == 0:
            print(json.dumps({"metric": "sparse Cholesky factor+solve GF/s", "value": work * args.steps / dt / 1e9, "unit": "GF/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "5-pt Laplacian %dx%d, n=%d, lower CCS int64, nrhs=%d, factor+solve per step" % (g, g, n, nrhs),
                                         "parallelism": "subtree-sharded x%d (cut depth %d, %.1f MB all-reduced per factorisation)"
                                                        % (world, DF.cut, DF.ulen * 8 / 1e6)},
                              "rel_residual": relres, "roofline": None, "cpu_baseline": None}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.quick:
        if rank == 0:
            print(json.dumps({"value": work * args.steps * world / dt / 1e9, "ms_per_step": dt / args.steps * 1e3, "ms_factor": ms_factor,
                              "ms_solve": ms_solve, "rel_residual": relres, "nsuper": int(info["nsuper"]), "nlevels": int(info["nlevels"]),
                              "lsize": int(info["lsize"]), "opts": args.chol_opts}))
        return
    dom = max(fam_times, key=lambda f: fam_times[f][0]) if args.roofline_family == "auto" else args.roofline_family
    dom_ms, dom_launches = fam_times[dom]
    if dom == "syrk_trailing":
        achieved = st["flops_syrk"] / (dom_ms * 1e-3) / 1e12
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
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01b_pmc_fetch_write_per_kernel.json")))
        if g == 1000 and nrhs == 1 and dom in pmc.get("family_bytes_per_step", {}):
            roofline["traffic"] = pmc["family_bytes_per_step"][dom] / max(dom_launches, 1)
            roofline["traffic_unit"] = "bytes per launch (FETCH_SIZE+WRITE_SIZE, raw; profiles/r01b_pmc_fetch_write_per_kernel.json)"
            roofline["algorithmic_bytes_per_launch"] = (alg_bytes / max(dom_launches, 1)) if roofline["bound"] == "hbm" else None
    except Exception:
        pass
    roofline["ms_per_step"] = dom_ms
    roofline["launches_per_step"] = dom_launches
    roofline["family_ms_per_step"] = {f: round(v[0], 4) for f, v in fam_times.items()}

    # --- CPU baseline: the oracle (plain-C restatement, 1 thread) on the same workload -----------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle.kvx_oracle import OracleChol
        perm = F.perm()
        O = OracleChol(n, colptr, rowind, "L", perm)
        tc0 = time.perf_counter()
        O.factorize(values)
        xb = np.asfortranarray(b_host.reshape(n, nrhs, order="F").copy())
        O.solve(xb)
        tc = time.perf_counter() - tc0
        ref_diff = float(np.abs(xb.reshape(-1, order="F") - x).max() / np.abs(xb).max())
        cpu = {"value": work / tc / 1e9, "unit": "GF/s", "cores": 1, "kind": "port",
               "sample": "same system and permutation, 1 numeric factorisation + 1 solve (%.1f s)" % tc,
               "max_rel_diff_vs_gpu": ref_diff}

    # --- the other half of BASELINE.json's metric: IPM iterations/s of the device-resident conelp on configs[3]
    # (inequality form), rank 0 only, a few hundred ms; never part of `value`
    ipm = None
    if rank == 0 and not args.no_ipm:
        try:
            from kvxopt_amd import lp as kvx_lp
            from kvxopt_amd.base import spmatrix
            Pl = workloads.lp_grid(250, 200)
            Gl = spmatrix.from_ccs(Pl["ml"], Pl["n"], Pl["Gp"], Pl["Gi"], Pl["Gx"])
            kvx_lp.conelp(Pl["c"], Gl, Pl["h"], options={"maxiters": 2})
            tl0 = time.perf_counter()
            sl = kvx_lp.conelp(Pl["c"], Gl, Pl["h"])
            tl = time.perf_counter() - tl0
            ipm = {"metric": "IPM iterations/s", "value": sl["iterations"] / sl["loop seconds"], "unit": "iterations/s",
                   "workload": "conelp, grid LP 250x200: ml=200000 inequalities, n=50000 (BASELINE configs[3], inequality form)",
                   "iterations": sl["iterations"], "status": sl["status"], "loop_s": sl["loop seconds"], "whole_call_s": tl,
                   "value_whole_call": sl["iterations"] / tl}
        except Exception as e:                      # the headline line must not depend on this leg
            ipm = {"error": repr(e)}

    if rank == 0:
        total = work * args.steps * world
        out = {
            "metric": "sparse Cholesky factor+solve GF/s", "value": total / dt / 1e9, "unit": "GF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "5-pt Laplacian %dx%d, n=%d, lower CCS int64, nrhs=%d, factor+solve per step" % (g, g, n, nrhs),
                       "nnz_lower": int(len(values)), "lnz": int(info["lnz"]), "flops_sum_cj2": info["flops"],
                       "nsuper": int(info["nsuper"]), "nlevels": int(info["nlevels"]), "max_front": int(info["max_front"]),
                       "analyze_s": round(t_analyze, 3), "parallelism": "replicas x%d" % world},
            "ms_factor": ms_factor, "ms_solve": ms_solve, "rel_residual": relres,
            "roofline": roofline, "cpu_baseline": cpu, "ipm": ipm,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()                  # rank 0 has extra legs (CPU baseline, IPM): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
