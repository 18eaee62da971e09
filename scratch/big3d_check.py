#!/usr/bin/env python3
"""Robustness check on larger / denser fronts: 3-D Laplacians (residual + timing), one GPU."""
import sys, time, json
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
for g in [int(a) for a in sys.argv[1:]] or [40, 64]:
    n, cp, ri, vx = workloads.laplacian_3d(g)
    t0 = time.time(); F = Factor(n, cp, ri); ta = time.time() - t0
    info = F.info()
    dev = torch.device('cuda', 0)
    v_d = torch.from_numpy(vx).to(dev)
    b = np.random.default_rng(5).standard_normal(n)
    x_d = torch.from_numpy(b.copy()).to(dev)
    F.factorize_dev(v_d.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    F.factorize_dev(v_d.data_ptr()); tf = time.perf_counter() - t0
    t0 = time.perf_counter(); F.solve_dev(x_d.data_ptr(), 0, 1, n); ts = time.perf_counter() - t0
    x = x_d.cpu().numpy()
    r = workloads.sym_matvec(n, cp, ri, vx, x.reshape(n, 1)) - b.reshape(n, 1)
    print(json.dumps({"grid": g, "n": n, "lnz": int(info["lnz"]), "flops": info["flops"], "max_front": int(info["max_front"]),
                      "nlevels": int(info["nlevels"]), "analyze_s": round(ta, 2), "factor_ms": tf * 1e3, "solve_ms": ts * 1e3,
                      "TF_s": info["flops"] / tf / 1e12, "rel_residual": float(np.linalg.norm(r) / np.linalg.norm(b))}), flush=True)
