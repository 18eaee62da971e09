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

One JSON line on stdout (rank 0).

Ranks.  `python bench.py --gpus N` with no torchrun environment starts the N ranks itself: the parent process -- before it
imports torch or touches HIP -- starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, lets
rank 0's line through and exits non-zero when any child does.  Under `python -m torch.distributed.run ... bench.py --gpus N`
the environment is already there and is used as it is.  N > 1 shards ONE system -- the SAME workload as N = 1 -- over the ranks
(`--dist subtree`, kvxopt_amd/dist.py: subtrees of the elimination tree by proportional mapping, block-cyclic top fronts with
panel broadcasts over RCCL) -> "scaling": "strong"; `--dist replicas` (every rank its own system, no data-path collective,
"weak") stays available.  `--workload lap3d --grid 200` is BASELINE configs[4].

The line carries `roofline` (dominant kernel family, timed live with HIP events on the stream it runs on), `cpu_baseline` (host
supernodal restatement on the host cores that go with one GPU and on 32 threads, SciPy SuperLU and the simplicial oracle
beside it), `ipm` (IPM iterations/s with its own roofline and CPU baseline), `extra` (the north-star systems measured in the
same run: 21-point stencil 1000^2 and the 7-point Laplacian 100^3, each with its own roofline and CPU baseline), `one_shot`
(time to first solution of cholmod.linsolve on a new pattern with host buffers, next to SciPy's splu) and `ranks` (backend,
world size, device bytes per rank).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The CPU baseline runs 16 OpenMP threads under a cgroup quota of 16 CPUs: threads that spin at the end of every parallel region
# burn the quota the working threads need (measured on the GPU box: 23.6 GF/s with the default policy, 34.1 with passive waits).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_MFMA_PEAK_TF = 78.6     # MI355X FP64 matrix = vector peak; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs; without a torchrun environment this process starts them itself")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, default=0, help="grid side (default: 1000 for the 5-pt Laplacian of config 2, 100 for --workload lap3d)")
    ap.add_argument("--workload", default="lap2d", choices=["lap2d", "lap3d", "stencil21"],
                    help="lap2d = BASELINE configs[1] (5-pt Laplacian, the headline); lap3d = 7-pt Laplacian on a cube (configs[4] is --grid 200); "
                         "stencil21 = the ~20 nnz/row system of north_star")
    ap.add_argument("--dist-ob", type=int, default=0, help="--dist subtree: column-block width of the block-cyclic fronts (0 = library default)")
    ap.add_argument("--dist-min-m", type=int, default=0, help="--dist subtree: smallest order of a block-cyclic front (0 = library default)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the host supernodal baseline (0 = the cores that go with one GPU, and 32)")
    ap.add_argument("--no-splu", action="store_true", help="skip the SciPy SuperLU line of the CPU baseline (it takes ~2 s on config 2)")
    ap.add_argument("--nrhs", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ipm", action="store_true", help="skip the secondary IPM iterations/s measurement")
    ap.add_argument("--no-klu", action="store_true", help="skip the klu leg (BASELINE configs[2]: ACTIVSg2000 and the 600 x 600 convection-diffusion matrix)")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` systems (21-point stencil, 100^3 cube)")
    ap.add_argument("--no-config5", action="store_true", help="skip the 200^3 cube (BASELINE configs[4]) on one GPU among the `extra` systems")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the time-to-first-solution leg")
    ap.add_argument("--separate-calls", action="store_true", help="the step as two calls (factorize, then solve) instead of the one-enqueue "
                    "form whose forward sweep is pipelined behind the factorisation")
    ap.add_argument("--roofline-family", default="auto")
    ap.add_argument("--chol-opts", default="", help="JSON dict of analysis options (nd_leaf, leaf_cols, leaf_rows, relax_*) for experiments")
    ap.add_argument("--quick", action="store_true", help="skip the per-family roofline loop, the CPU baselines and the secondary legs (experiments)")
    ap.add_argument("--dist", default=None, choices=["replicas", "subtree"],
                    help="N > 1 ranks: 'subtree' (default) = ONE system -- the N = 1 workload -- sharded over the ranks (kvxopt_amd.dist.DistFactor: "
                         "subtrees by proportional mapping, block-cyclic top fronts with panel broadcasts; strong scaling); "
                         "'replicas' = every rank factors its own system (weak scaling, no data-path collective)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks from THIS process, which has not imported torch and has
    made no HIP call (a process that has initialised the GPU must never be replaced or forked into ranks).  Children inherit
    stdout / stderr -- only rank 0 prints the JSON line.  Returns the exit code: 0 when every rank ended with 0."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL between processes needs it on this host driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for pr in list(alive):
            code = pr.poll()
            if code is None:
                continue
            alive.remove(pr)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for other in alive:                                 # one rank failed: the others would wait in a collective for ever
                    other.terminate()
    if rc:
        sys.stderr.write("bench.py: a rank exited with code %d\n" % rc)
    return rc


class TorchRanks:
    """The ranks of the job through torch.distributed (KVX_DIST_BACKEND=nccl: RCCL through torch; gloo: rehearsal on fewer GPUs)."""

    def __init__(self, torch, dist, dev):
        self.torch, self.dist, self.dev = torch, dist, dev
        self.backend, self.world, self.comm = dist.get_backend(), dist.get_world_size(), None

    def barrier(self):
        self.dist.barrier()

    def max(self, v):
        t = self.torch.tensor([float(v)], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, values):
        mine = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.dev)
        got = [mine.clone() for _ in range(self.world)]
        self.dist.all_gather(got, mine)
        return [[float(x) for x in t.cpu().tolist()] for t in got]

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


class RcclRanks:
    """The ranks of the job through librccl.so itself (kvxopt_amd.rccl.World, csrc/rccl_comm.cpp): no torch in the process, the
    system's HIP runtime, collectives enqueued from C.  The default for N > 1."""

    def __init__(self, W):
        self.comm, self.backend, self.world = W, "rccl-direct (librccl %d)" % W.version, W.world

    def barrier(self):
        self.comm.barrier()

    def max(self, v):
        return self.comm.max(v)

    def gather(self, values):
        return [[float(x) for x in row] for row in self.comm.all_gather(values)]

    def close(self):
        self.comm.barrier()
        self.comm.close()


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


def cgroup_cpu_quota():
    """CPUs' worth of time the process's cgroup may use (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def host_cores():
    """Cores this process can keep busy: its affinity mask, capped by the cgroup's CPU quota (a GPU box hands one GPU's share of
    the node's cores to a job as a quota, not as a mask: 32 threads on a 16-CPU quota lose to 16 -- round 3's by_threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    if q is not None:
        n = max(1, min(n, int(q + 0.5)))
    return n


def host_threads(args):
    """Thread counts of the host supernodal baseline: the cores that go with ONE GPU of this node (16 of 256 on an 8-GPU box;
    fewer when the process may run on fewer) and 32 -- the most the OpenBLAS inside scipy takes (it is built for 64 threads and
    its buffer table overflows when more OpenMP threads than that call it at once).  --cpu-threads T: that count only."""
    avail = host_cores()
    if args.cpu_threads:
        return [max(1, min(int(args.cpu_threads), 32, avail))]
    out = [max(1, min(avail, 16))]
    if avail >= 32:
        out.append(32)
    return out


def cpu_baselines(args, F, n, colptr, rowind, values, b_host, nrhs, work, x_gpu, lite=False):
    """The reference's CPU path is SuiteSparse CHOLMOD (third-party, absent from this image: probed below).  Timed instead, on
    the same matrix, permutation and flop count: (1) the host supernodal multifrontal restatement (OpenMP subtrees + OpenBLAS
    BLAS-3 in the fronts, oracle/kvx_supernodal.c) on the cores that go with one GPU AND on 32 threads -- the better of the two
    is the headline baseline, `cores` says which it was and `cores_available` what the box has; (2) SciPy SuperLU on P A P' with
    the natural column order, 1 thread (the calibration of BASELINE.md section 2); (3) the simplicial up-looking oracle, 1 core
    (the parity checker).  lite: (1) only, one repetition per thread count (the `extra` systems)."""
    import ctypes.util
    B = np.asfortranarray(b_host.reshape(n, nrhs, order="F").copy())
    out = {"cholmod_found": bool(ctypes.util.find_library("cholmod")), "os_cpu_count": os.cpu_count(), "cores_available": host_cores(), "cgroup_cpu_quota": cgroup_cpu_quota(),
           "env_threads": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS") if os.environ.get(k)}}
    others = {}
    runs = []
    try:
        from oracle.kvx_oracle import OracleSupernodal
        for cores in host_threads(args):
            S = OracleSupernodal.from_factor(n, colptr, rowind, F, threads=cores)
            best = None
            for _ in range(2 if lite else 3):               # first pass pays page faults and thread start-up: best of three (two on the `extra` systems)
                xb = B.copy(order="F")
                t0 = time.perf_counter(); S.factorize(values); t1 = time.perf_counter(); S.solve(xb); t2 = time.perf_counter()
                if best is None or t2 - t0 < best[0]:
                    best = (t2 - t0, t1 - t0, t2 - t1)
            diff = float(np.abs(xb.reshape(-1, order="F") - x_gpu).max() / np.abs(xb).max())
            runs.append({"cores": cores, "value": work / best[0] / 1e9, "factor_s": best[1], "solve_s": best[2], "max_rel_diff_vs_gpu": diff})
            del S
        top = max(runs, key=lambda r: r["value"])
        out.update({"value": top["value"], "unit": "GF/s", "cores": top["cores"], "kind": "port",
                    "library": "host supernodal multifrontal restatement (oracle/kvx_supernodal.c): OpenMP over elimination-tree subtrees, "
                               "OpenBLAS dpotrf/dtrsm/dsyrk inside the fronts (the OpenBLAS that ships in scipy)",
                    "sample": "same system, permutation and supernodes as the GPU run: numeric factorisation %.3f s + solve %.3f s on %d threads, best of %d"
                              % (top["factor_s"], top["solve_s"], top["cores"], 2 if lite else 3),
                    "max_rel_diff_vs_gpu": top["max_rel_diff_vs_gpu"], "by_threads": runs})
    except Exception as e:
        out["supernodal_error"] = repr(e)
    if lite:
        return out
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


IPM_PMC_FILE = "r04_ipm_pmc.json"


def ipm_leg(args, pl):
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
    gx = pl.to_dev(Pl["Gx"]); w = pl.to_dev(np.random.default_rng(0).uniform(0.5, 1.5, ml))
    sx = pl.empty(max(snz.value, 1))
    reps = 50
    for _ in range(5):
        _lib.raise_for(L.kvx_atda_assemble_dev(h, pl.ptr(gx), pl.ptr(w), None, pl.ptr(sx)))
    pl.sync()
    t0 = time.perf_counter()                      # back-to-back launches on the null stream between two device synchronisations
    for _ in range(reps):
        _lib.raise_for(L.kvx_atda_assemble_dev(h, pl.ptr(gx), pl.ptr(w), None, pl.ptr(sx)))
    pl.sync()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    alg = 12.0 * len(Pl["Gx"]) + 8.0 * ml + 8.0 * snz.value
    ipm["roofline"] = {"kernel": "k_atda_scale + k_atda (S = G' diag(w) G on the fixed pattern: two launches)", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_launch": alg, "ms_per_launch": ms, "launches_timed": reps,
                       "note": "back-to-back launches: time per launch includes the launch gap; the loop itself is bound by the factor/solve latency chains"}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", IPM_PMC_FILE)))
        ipm["roofline"]["traffic"] = pm.get("k_atda_bytes_per_launch")
    except Exception:
        pass
    # The family that DOMINATES an iteration is not the assembly (0.3 % of the GPU time) but the factorisation of S and its
    # solves: the same per-family timing as the headline's roofline leg, on S itself (pattern of G'G from the plan, values of the
    # assembly above), one factorisation + a two-column solve per step as an iteration's first direction does.
    ipm["roofline_assembly"] = ipm["roofline"]
    try:
        from kvxopt_amd.chol import Factor
        Sp = np.zeros(nl + 1, dtype=np.int64); Si = np.zeros(max(snz.value, 1), dtype=np.int64)
        _lib.raise_for(L.kvx_atda_pattern(h, ctypes.byref(snz), _lib.pi(Sp), _lib.pi(Si)))
        FS = Factor(nl, Sp, Si[:snz.value], "L", None, None)
        rhs = pl.to_dev(np.random.default_rng(1).standard_normal(2 * nl))
        xs = pl.empty(2 * nl)

        def step_s():
            pl.copy(xs, rhs, 2 * nl)
            FS.factorize_solve_dev(pl.ptr(sx), pl.ptr(xs), 2, nl)
        for _ in range(4):
            step_s()
        stS = front_stats(FS)
        famS = {}
        for fam in Factor.FAMILIES:
            smp = []
            for _ in range(3):
                FS.prof_select(fam); step_s(); smp.append(FS.prof_read())
            smp.sort(key=lambda t: t[0])
            famS[fam] = smp[1]
        FS.prof_select(None)
        dom = max(famS, key=lambda f: famS[f][0])
        dom_ms, dom_l = famS[dom]
        if dom == "syrk_trailing":
            ach = stS["flops_syrk"] / (dom_ms * 1e-3) / 1e12
            rf = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TF}
        else:
            ab = stS["bytes_small"] if dom == "front_small" else ((8.0 * stS["lsize"] + 4.0 * stS["sum_m"] + 16.0 * nl) * 2 if dom in ("fwd_level", "bwd_level")
                                                                   else 8.0 * stS["lsize"])
            ach = ab / (dom_ms * 1e-3) / 1e9
            rf = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_step": ab}
        rf.update({"traffic": None, "ms_per_step": dom_ms, "launches_per_step": dom_l,
                   "family_ms_per_step": {f: round(v[0], 4) for f, v in famS.items()},
                   "what": "dominant kernel family of the factorisation of S (order %d, nnz(L) %d, %d fronts in %d levels) + two-column solve that "
                           "an iteration's first direction enqueues; HIP events around every launch of the family, graphs off for these passes"
                           % (nl, int(FS.info()["lnz"]), int(FS.info()["nsuper"]), int(FS.info()["nlevels"]))})
        ipm["roofline"] = rf
        del FS
    except Exception as e:
        ipm["roofline_dominant_error"] = repr(e)
    L.kvx_atda_free(h)
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
            cores = host_threads(args)[0]
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


def source_fingerprint():
    """sha256 over the library sources (csrc/*): a committed PMC profile names the sources it was taken on, and the line says
    when the kernels have changed since (`traffic_stale`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "kvxopt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "kvxopt_amd", "csrc", "*.[ch]pp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_FILE = "r04_pmc_fetch_write_per_kernel.json"


def build_workload(name, g):
    from kvxopt_amd import workloads
    if name == "lap2d":
        return workloads.laplacian_2d(g) + ("5-pt Laplacian %dx%d" % (g, g),)
    if name == "lap3d":
        return workloads.laplacian_3d(g) + ("7-pt Laplacian %dx%dx%d" % (g, g, g),)
    return workloads.stencil21_2d(g) + ("21-pt stencil %dx%d (radius-2 box+cross, ~20 nnz/row, seed 20)" % (g, g),)


class Plumbing:
    """Device memory, copies and synchronisation of one rank.  A run with ranks needs torch.distributed, so there the buffers are
    torch tensors (torch's default stream IS the null stream the library orders itself against).  A single-GPU run does not
    import torch at all: buffers come from the library's own allocator (kvx_dev_malloc) and the process runs on the system's HIP
    runtime (ROCm 7.2) instead of the 7.0.51831 bundled with the PyTorch wheel -- under which the one-enqueue step falls back to
    its two separate graphs (include/kvxhip.h, kvx_chol_factorize_solve_dev)."""

    def __init__(self, torch=None, dev=None):
        self.torch, self.dev = torch, dev

    def to_dev(self, a):
        a = np.ascontiguousarray(a)
        if self.torch is not None:
            return self.torch.from_numpy(a).to(self.dev)
        from kvxopt_amd import _lib
        return _lib.DeviceBuffer.from_array(a)

    def empty(self, ndoubles):
        if self.torch is not None:
            return self.torch.empty(ndoubles, dtype=self.torch.float64, device=self.dev)
        from kvxopt_amd import _lib
        return _lib.DeviceBuffer(8 * ndoubles)

    def ptr(self, h):
        return h.data_ptr() if self.torch is not None else h.ptr

    def copy(self, dst, src, ndoubles):                      # on the null stream
        if self.torch is not None:
            dst.copy_(src)
        else:
            from kvxopt_amd import _lib
            _lib.raise_for(_lib.lib().kvx_vec_copy_dev(ndoubles, src.ptr, dst.ptr))

    def to_host(self, h, ndoubles):
        return h.cpu().numpy() if self.torch is not None else h.download(np.float64, ndoubles)

    def sync(self):
        if self.torch is not None:
            self.torch.cuda.synchronize()
        else:
            from kvxopt_amd import _lib
            _lib.raise_for(_lib.lib().kvx_dev_sync())

    def mem_info(self):
        from kvxopt_amd import _lib
        f, t = ctypes.c_int64(), ctypes.c_int64()
        _lib.raise_for(_lib.lib().kvx_dev_mem_info(ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def runtime(self):
        return ("HIP runtime of the PyTorch wheel %s" % self.torch.version.hip) if self.torch is not None else "system HIP runtime (/opt/rocm), torch not imported"


def measure_system(args, pl, dist, rank, world, dev, wl, nrhs, steps, warmup, mode, family="auto", cpu="full", pmc_key=None):
    """Factor + solve of one system, timed as the contract says (barrier + synchronize on both sides, MAX over the ranks),
    with the roofline of the dominant kernel family (HIP events on the factor's own streams) and the CPU baselines beside it.
    mode: 'single' (this rank alone), 'replicas' (every rank its own copy), 'subtree' (ONE system over the ranks).
    Returns the result dict on every rank (CPU legs on rank 0 only)."""
    from kvxopt_amd import workloads
    from kvxopt_amd.chol import Factor
    n, colptr, rowind, values, wname = wl
    chol_opts = json.loads(args.chol_opts) if args.chol_opts else None
    DF = None
    t0 = time.time()
    if mode == "subtree":
        from kvxopt_amd.dist import DistFactor
        DF = DistFactor(n, colptr, rowind, "L", None, chol_opts, device=dev, ob=args.dist_ob, min_m=args.dist_min_m,
                        comm=(dist.comm if dist is not None else None))
        F = DF.F
    else:
        F = Factor(n, colptr, rowind, "L", None, chol_opts)
    t_analyze = time.time() - t0
    info = F.info()
    work = info["flops"] + 4.0 * info["lnz"] * nrhs          # SURVEY 8(d) flop measure per step
    torch = pl.torch
    vals_d = pl.to_dev(values)
    b_host = np.random.default_rng(2).standard_normal((n, nrhs)).reshape(n * nrhs, order="F") if nrhs > 1 \
        else np.random.default_rng(2).standard_normal(n)
    b_d = pl.to_dev(b_host)
    x_d = pl.empty(n * nrhs)
    pl.sync()

    def step():
        if DF is not None:                                   # one system over all ranks
            DF.factorize(vals_d)
            pl.copy(x_d, b_d, n * nrhs)
            DF.solve(x_d, nrhs)
            return
        if args.separate_calls:                              # numeric() then solve(): two entry points, the sweep waits for the whole factor
            F.factorize_dev(pl.ptr(vals_d), sync=False)
            pl.copy(x_d, b_d, n * nrhs)
            F.solve_dev(pl.ptr(x_d), 0, nrhs, n)             # ordered behind the copy (null stream) by an event; synchronises the factor's stream
            return
        # numeric() + solve() as ONE enqueue (kvx_chol_factorize_solve_dev): the forward sweep follows the factorisation level by
        # level on its own streams -- same kernels, same order per front, bitwise the same factor and solution as the two calls
        pl.copy(x_d, b_d, n * nrhs)
        F.factorize_solve_dev(pl.ptr(vals_d), pl.ptr(x_d), nrhs, n)   # ordered behind the copy by an event; synchronises

    collective = dist is not None and mode != "single"

    def barrier():
        if collective:
            dist.barrier()
        pl.sync()

    for _ in range(warmup):
        step()
    # --- timed region ------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if collective:
        dt = dist.max(dt)
    ms_factor, ms_solve = F.timing()
    ms_sep = None
    heavy = info["flops"] > 2e13                             # (the 200^3 cube: seconds per step -- no split, one reading of one family)
    if DF is None and not args.separate_calls and not heavy:
        # the one-enqueue step cannot tell its two parts apart: a few steps as two calls (factorize, then solve) for the split
        acc = [0.0, 0.0]
        for it in range(8):                                  # (three to capture and warm the two graphs of this form, five measured)
            F.factorize_dev(pl.ptr(vals_d), sync=False)
            pl.copy(x_d, b_d, n * nrhs)
            F.solve_dev(pl.ptr(x_d), 0, nrhs, n)
            a, b_ = F.timing()
            if it >= 3:
                acc[0] += a / 5; acc[1] += b_ / 5
        ms_sep = {"ms_factor": acc[0], "ms_solve": acc[1], "note": "the same work as two calls (factorize, then solve), 5 steps outside the timed region"}
    steps_run = warmup + steps
    coll = (DF.collectives, DF.bytes_moved) if DF is not None else (0, 0)   # (counted up to here: warm-up + timed steps)

    # residual of the last solve (parity bar: <= 1e-10 relative)
    x = pl.to_host(x_d, n * nrhs)
    r = workloads.sym_matvec(n, colptr, rowind, values, x.reshape(n, nrhs, order="F")) - b_host.reshape(n, nrhs, order="F")
    relres = float(np.linalg.norm(r) / np.linalg.norm(b_host))
    nsys = world if mode == "replicas" else 1
    res = {"workload": wname, "n": int(n), "nrhs": nrhs, "value": work * steps * nsys / dt / 1e9, "unit": "GF/s", "ms_per_step": dt / steps * 1e3,
           "ms_factor": ms_factor, "ms_solve": ms_solve, "as_two_calls": ms_sep, "rel_residual": relres, "steps": steps, "warmup": warmup,
           "step_form": ("two calls: factorize, then solve" if (args.separate_calls or DF is not None) else
                         "one enqueue (kvx_chol_factorize_solve_dev): the forward sweep runs beside the factorisation of the top of the tree; "
                         "ms_factor is the whole call and ms_solve 0 -- the two parts are not separable (see as_two_calls)"),
           "nnz_lower": int(len(values)), "lnz": int(info["lnz"]), "flops_sum_cj2": info["flops"], "nsuper": int(info["nsuper"]),
           "nlevels": int(info["nlevels"]), "max_front": int(info["max_front"]), "lsize": int(info["lsize"]), "analyze_s": round(t_analyze, 3),
           "dev_bytes_rank0": int(F.info()["dev_bytes"]), "_work": work, "_dt": dt, "_x": x, "_F": F, "_DF": DF}
    if DF is not None:
        res["sharding"] = {"shared_fronts_rank0": DF.nshared, "block_cyclic_rank0": DF.ncyclic, "block_columns": DF.ob,
                           "collectives_per_step_rank0": coll[0] // max(steps_run, 1), "mb_moved_per_step_rank0": coll[1] / max(steps_run, 1) / 1e6,
                           "panel_doubles_rank0": DF.lsize_local, "panel_doubles_total": DF.lsize_total}
    if args.quick:
        return res

    # --- roofline leg: dominant kernel family timed live with HIP events on the factor's stream (hipGraph replay is off for
    # these passes: the events go around the individual launches) -----------------------------------------------------------
    st = front_stats(F)
    fam_times = {}
    for fam in Factor.FAMILIES:
        if heavy and fam != (family if family != "auto" else "syrk_trailing"):
            fam_times[fam] = (0.0, 0)
            continue
        samples = []
        for _ in range(1 if heavy else 3):         # median of three single-step readings: one hiccup must not pick the family
            F.prof_select(fam)
            step()
            samples.append(F.prof_read())
        samples.sort(key=lambda t: t[0])
        fam_times[fam] = (samples[len(samples) // 2][0], samples[len(samples) // 2][1])
    F.prof_select(None)
    pick = args.roofline_family if family == "auto" else family
    dom = max(fam_times, key=lambda f: fam_times[f][0]) if pick == "auto" else pick
    dom_ms, dom_launches = fam_times[dom]
    if DF is not None and world > 1:
        # rank 0 times its own launches: scale the family's work by the share of the flops rank 0 executes (host-side map)
        from kvxopt_amd.dist import partition
        mp = partition(F, world, DF.ob, DF.min_m)              # (the map reads the tree and the front sizes only: the per-rank layout does not change it)
        share0 = float(mp["rank_flops"][0] / mp["flops"])
        res["sharding"]["flop_share_by_rank"] = [round(float(v / mp["flops"]), 4) for v in mp["rank_flops"]]
        st = {k: (v * share0 if isinstance(v, float) else v) for k, v in st.items()}
    alg_bytes = None
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
            alg_bytes = 24.0 * len(values) + 8.0 * st["lsize"]      # (the family zeroes L in the same pass since round 3)
        else:
            alg_bytes = 8.0 * st["lsize"]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None}
    roofline["timing_mode"] = "HIP events around every launch of the family on the stream it runs on, hipGraph replay off for these passes (the timed step replays graphs)"
    # HBM traffic of the dominant family from the committed PMC passes (profiles/, separate rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE runs of this same command; not measured in this run); null when no profile matches this workload,
    # `traffic_stale` when the library sources have changed since the profile was taken.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        fam_b = pmc.get(pmc_key or "", {}).get("family_bytes_per_step", {}) if pmc_key else {}
        if DF is None and dom in fam_b:
            roofline["traffic"] = fam_b[dom] / max(dom_launches, 1)
            roofline["traffic_unit"] = "bytes per launch, FETCH_SIZE x fetch_scale + WRITE_SIZE, from the committed profile profiles/%s (fetch_scale %s: %s)" % (
                PMC_FILE, pmc.get("fetch_scale"), pmc.get("fetch_scale_note", ""))
            roofline["traffic_stale"] = pmc.get("source_fingerprint") != source_fingerprint()
            roofline["algorithmic_bytes_per_launch"] = (alg_bytes / max(dom_launches, 1)) if alg_bytes is not None else None
    except Exception:
        pass
    roofline["ms_per_step"] = dom_ms
    roofline["launches_per_step"] = dom_launches
    roofline["family_ms_per_step"] = {f: round(v[0], 4) for f, v in fam_times.items()}
    roofline["also"] = {"flops_syrk_per_step": st["flops_syrk"], "mfma_frac_syrk_trailing": (st["flops_syrk"] / (fam_times["syrk_trailing"][0] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF) if fam_times["syrk_trailing"][0] > 0 else None,
                        "bytes_small_per_step": st["bytes_small"], "hbm_frac_front_small": (st["bytes_small"] / (fam_times["front_small"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if fam_times["front_small"][0] > 0 else None}
    res["roofline"] = roofline

    # --- CPU baselines on the same system and permutation (rank 0, host cores of this box) -------
    res["cpu_baseline"] = None
    if rank == 0 and cpu != "none" and not args.no_cpu_baseline and (world == 1 or mode != "replicas"):
        res["cpu_baseline"] = cpu_baselines(args, F, n, colptr, rowind, values, b_host, nrhs, work, x, lite=(cpu == "lite"))
    return res


def public(res):
    return {k: v for k, v in res.items() if not k.startswith("_")}


def klu_leg(args):
    """BASELINE configs[2]: kvxopt.klu on ACTIVSg2000 (refactor + solve on the GPU, klu.linsolve with host buffers, SciPy SuperLU
    beside it) and the blocked path on a 600 x 600 convection-diffusion matrix, each with the roofline of a refactorisation:
    algorithmic bytes 8 (panel entries + 2 update entries) + 12 nnz(A) (kvx_lu_num_work) for the latency-bound small case, flops
    against the FP64 MFMA peak for the blocked one."""
    import bench_extra
    from kvxopt_amd import klu as kvx_klu
    from kvxopt_amd.base import spmatrix
    out = {}
    r = bench_extra.klu_case(20, 5)
    z = np.load(os.path.join(ROOT, "tests", "golden", "ACTIVSg2000.npz"))
    A = spmatrix.from_ccs(int(z["n"]), int(z["n"]), z["colptr"], z["rowind"], z["values"])
    Fn = kvx_klu.numeric(A, kvx_klu.symbolic(A))
    w = Fn.num.work()
    ab = 8.0 * (w["panel_entries"] + 2.0 * w["update_entries"]) + 12.0 * r["nnz"]
    ach = ab / (r["ms_refactor_dev"] * 1e-3) / 1e9
    r["roofline"] = {"kernel": "LU refactorisation, all launches (k_lu_* one-workgroup fronts: %d of %d fronts blocked)" % (int(w["blocked_fronts"]), r["lu_nfront"]),
                     "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes": ab, "flops": w["flops"], "ms": r["ms_refactor_dev"],
                     "note": "%d fronts in %d dependent levels: latency-bound, as the number says" % (r["lu_nfront"], r["lu_nlevels"])}
    out["activsg2000"] = r
    try:
        g = 600
        r2 = bench_extra.lu2d_case(g, 5)
        from kvxopt_amd import workloads
        n2, cp2, ri2, v2 = workloads.convdiff_2d(g)
        A2 = spmatrix.from_ccs(n2, n2, cp2, ri2, v2)
        F2 = kvx_klu.numeric(A2, kvx_klu.symbolic(A2))
        w2 = F2.num.work()
        ach2 = w2["flops"] / (r2["ms_refactor_dev"] * 1e-3) / 1e12
        r2["roofline"] = {"kernel": "LU refactorisation, all launches (k_lub_* blocked fronts carry %.0f %% of the flops)" % (100.0 * w2["blocked_flops"] / max(w2["flops"], 1.0)),
                          "bound": "mfma", "achieved": ach2, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach2 / FP64_MFMA_PEAK_TF,
                          "traffic": None, "flops": w2["flops"], "ms": r2["ms_refactor_dev"],
                          "algorithmic_bytes": 8.0 * (w2["panel_entries"] + 2.0 * w2["update_entries"]) + 12.0 * r2["nnz"]}
        del F2
        out["convdiff600"] = r2
    except Exception as e:
        out["convdiff600"] = {"error": repr(e)}
    return out


def one_shot_leg(args, wl):
    """Time to first solution on a NEW pattern with HOST buffers -- what one cholmod.linsolve call costs (cholmod.c:618-753:
    analysis + upload + numeric factorisation + solve + download) -- next to SciPy's splu (ordering + factorisation + solve) on
    this box's host.  The object cache is emptied first; the library itself (code objects, pools) is warm."""
    from kvxopt_amd import cholmod
    from kvxopt_amd.base import matrix, spmatrix
    n, colptr, rowind, values, wname = wl
    A = spmatrix.from_ccs(n, n, colptr, rowind, values)
    b = np.random.default_rng(2).standard_normal(n)
    cholmod.clear_cache()
    B = matrix(b.copy())
    t0 = time.perf_counter()
    cholmod.linsolve(A, B)
    t_first = time.perf_counter() - t0
    x = np.asarray(B._a).reshape(-1).copy()
    B2 = matrix(b.copy())
    t0 = time.perf_counter()
    cholmod.linsolve(A, B2)                                  # the same pattern again: refactorisation + solve on the kept analysis
    t_again = time.perf_counter() - t0
    cholmod.clear_cache()
    out = {"workload": wname + ", cholmod.linsolve(A, b) with host buffers", "gpu_first_call_s": t_first, "gpu_known_pattern_s": t_again,
           "measured": "wall time of the call: host analysis (ordering, symbolic), upload, numeric factorisation, solve, download"}
    if not args.no_splu:
        try:
            import scipy.sparse as sp
            from scipy.sparse.linalg import splu
            Al = sp.csc_matrix((values, rowind, colptr), shape=(n, n))
            Af = (Al + sp.tril(Al, -1).T).tocsc()
            t0 = time.perf_counter()
            lu = splu(Af, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options={"SymmetricMode": True})
            xs = lu.solve(b)
            t_cpu = time.perf_counter() - t0
            out["scipy_splu_s"] = t_cpu
            out["scipy_splu"] = "splu(A, permc_spec='MMD_AT_PLUS_A', SymmetricMode) + solve, 1 thread (the calibration set-up of BASELINE.md section 2)"
            out["speedup_first_call"] = t_cpu / t_first
            out["max_rel_diff_vs_splu"] = float(np.abs(xs - x).max() / np.abs(xs).max())
        except Exception as e:
            out["scipy_splu_error"] = repr(e)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))                          # (nothing above has imported torch or touched HIP)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from kvxopt_amd import _lib as _kvx_lib
    dist = None
    if world == 1:
        # one GPU: no torch in the process at all (Plumbing): the library's own allocator and the system's HIP runtime
        try:
            _kvx_lib.require_device()
        except Exception:
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        pl = Plumbing()
        dev = None
    else:
        # one GPU per rank.  KVX_DIST_BACKEND: "rccl" (default) = librccl.so bound directly, no torch in the process; "nccl" = RCCL
        # through torch.distributed; "gloo" rehearses the N > 1 paths on a box with fewer GPUs than ranks (torch, host-staged)
        backend = os.environ.get("KVX_DIST_BACKEND", "rccl")
        if backend == "rccl":
            try:
                _kvx_lib.require_device()
            except Exception:
                raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
            ndev = int(_kvx_lib.lib().kvx_device_count())
            if world > ndev:
                raise SystemExit("bench.py: %d ranks but %d visible GPU(s); RCCL needs one GPU per rank (KVX_DIST_BACKEND=gloo rehearses the sharded path on fewer)" % (world, ndev))
            from kvxopt_amd.rccl import World
            dist = RcclRanks(World(rank, world, local_rank))
            pl = Plumbing()
            dev = None
            assert "torch" not in sys.modules
        else:
            import torch
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
            ndev = max(torch.cuda.device_count(), 1)
            if backend != "nccl":
                local_rank = local_rank % ndev
            elif world > ndev:
                raise SystemExit("bench.py: %d ranks but %d visible GPU(s); RCCL needs one GPU per rank (KVX_DIST_BACKEND=gloo rehearses the sharded path on fewer)" % (world, ndev))
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            pl = Plumbing(torch, dev)
            import torch.distributed as dist_
            if backend == "nccl":
                dist_.init_process_group("nccl", device_id=dev)
            else:
                dist_.init_process_group(backend)
            dist = TorchRanks(torch, dist_, dev)
    mode = "single" if world == 1 else (args.dist or "subtree")

    g = args.grid or (100 if args.workload == "lap3d" else 1000)
    wl = build_workload(args.workload, g)
    headline_cfg2 = args.workload == "lap2d" and g == 1000 and args.nrhs == 1
    res = measure_system(args, pl, dist, rank, world, dev, wl, args.nrhs, args.steps, args.warmup, mode,
                         pmc_key=("config2" if headline_cfg2 and mode == "single" else None))
    if args.quick:
        if rank == 0:
            print(json.dumps(dict(public(res), opts=args.chol_opts, n_gpus=world)))
        if dist is not None:
            dist.close()
        return
    # per-rank device memory: the factor's own large buffers (library count) and what the device reports in use
    free_b, total_b = pl.mem_info()
    mine_l = [float(res["_F"].info()["dev_bytes"]), float(total_b - free_b)]
    per_rank = [mine_l]
    if dist is not None:
        per_rank = dist.gather(mine_l)
    ranks = {"backend": (dist.backend if dist is not None else "none"), "world_size": (dist.world if dist is not None else 1),
             "factor_bytes_by_rank": [int(t[0]) for t in per_rank], "device_bytes_in_use_by_rank": [int(t[1]) for t in per_rank],
             "hip_runtime": pl.runtime()}

    # --- the north-star systems in the same run (every N; CPU baselines at N = 1): ~20 nnz/row at n = 1e6, and the flop-bound cube
    # --- BASELINE configs[2] (kvxopt.klu), rank 0 only.  Before the `extra` systems: klu.linsolve is a 1 ms call bound by host
    # latency, and right after the 170 GB factor of the 200^3 system the same call measured 1.04 ms against 0.91 here
    klu = None
    if rank == 0 and world == 1 and not args.no_klu and headline_cfg2:
        try:
            klu = klu_leg(args)
        except Exception as e:
            klu = {"error": repr(e)}

    extra = []
    if not args.no_extra and headline_cfg2:
        systems = [("stencil21", 1000, 5, 2), ("lap3d", 100, 3, 1)]
        if world == 1 and not args.no_config5:
            # BASELINE configs[4] on ONE GPU: the N = 1 anchor of the 8-GPU configuration (n = 8e6; no CPU run: the restated CPU path
            # would take minutes).  Two steps; ~73 GB of factor.
            systems.append(("lap3d", 200, 1, 1))
        for name, gg, st_, wu in systems:
            try:
                del res["_F"], res["_DF"]                   # (one factor resident at a time)
            except KeyError:
                pass
            try:
                w2 = build_workload(name, gg)
                r2 = measure_system(args, pl, dist, rank, world, dev, w2, 1, st_, wu, mode, family="syrk_trailing",
                                    cpu=("lite" if (world == 1 and not (name == "lap3d" and gg == 200)) else "none"), pmc_key=("stencil21" if (name == "stencil21" and mode == "single") else None))
                if r2.get("cpu_baseline") and r2["cpu_baseline"].get("value"):
                    r2["vs_cpu_baseline"] = r2["value"] / r2["cpu_baseline"]["value"]
                extra.append(public(r2))
                del r2
            except Exception as e:                          # the headline line must not depend on these legs
                extra.append({"workload": name, "error": repr(e)})

    # --- the other half of BASELINE.json's metric: IPM iterations/s of the device-resident conelp on configs[3]
    # (inequality form), rank 0 only, a few hundred ms; never part of `value`
    ipm = None
    one_shot = None
    if rank == 0 and world == 1:
        if not args.no_ipm:
            try:
                ipm = ipm_leg(args, pl)
            except Exception as e:                      # the headline line must not depend on this leg
                ipm = {"error": repr(e)}
        if not args.no_one_shot and headline_cfg2:
            try:
                one_shot = one_shot_leg(args, wl)
            except Exception as e:
                one_shot = {"error": repr(e)}

    if rank == 0:
        cpu = res.get("cpu_baseline")
        n = res["n"]
        par = "1 GPU" if world == 1 else (("replicas x%d: every rank its own system, no data-path collective" % world) if mode == "replicas" else
                                         ("ONE system sharded x%d over %s: subtrees by proportional mapping, %d shared fronts on rank 0 (%d block-cyclic, "
                                          "%d-column blocks), %d collectives / %.1f MB per step on rank 0"
                                          % (world, ranks["backend"], res["sharding"]["shared_fronts_rank0"], res["sharding"]["block_cyclic_rank0"],
                                             res["sharding"]["block_columns"], res["sharding"]["collectives_per_step_rank0"],
                                             res["sharding"]["mb_moved_per_step_rank0"])))
        out = {
            "metric": "sparse Cholesky factor+solve GF/s", "value": res["value"], "unit": "GF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak" if mode == "replicas" else "strong",
            "vs_baseline": None,                    # BASELINE.md holds no published number for this metric
            "vs_cpu_baseline": (res["value"] / cpu["value"]) if (cpu and cpu.get("value")) else None,
            "vs_cpu_baseline_note": "GPU / the restated CPU path (host supernodal port, `cpu_baseline.cores` of `cpu_baseline.cores_available` cores); "
                                    "the reference's own CPU library (SuiteSparse CHOLMOD) is not on this box",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, n=%d, lower CCS int64, nrhs=%d, factor+solve per step" % (res["workload"], n, args.nrhs),
                       "nnz_lower": res["nnz_lower"], "lnz": res["lnz"], "flops_sum_cj2": res["flops_sum_cj2"],
                       "nsuper": res["nsuper"], "nlevels": res["nlevels"], "max_front": res["max_front"],
                       "analyze_s": res["analyze_s"], "parallelism": par},
            "ms_factor": res["ms_factor"], "ms_solve": res["ms_solve"], "as_two_calls": res.get("as_two_calls"), "step_form": res["step_form"],
            "rel_residual": res["rel_residual"],
            "roofline": res["roofline"], "cpu_baseline": cpu, "ranks": ranks, "sharding": res.get("sharding"),
            "extra": extra, "ipm": ipm, "klu": klu, "one_shot": one_shot,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.close()                    # (barrier first: rank 0 has extra legs -- CPU baseline, IPM -- leave together)


if __name__ == "__main__":
    main()
