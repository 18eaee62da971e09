#!/usr/bin/env python3
"""Dense SPD matrix as one front: isolates the big-front path (potrf/trsm/syrk) -- dense Cholesky TF/s."""
import sys, time, json
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from kvxopt_amd.chol import Factor
from kvxopt_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
cp = np.zeros(n + 1, dtype=np.int64)
cp[1:] = np.cumsum(np.arange(n, 0, -1))
ri = np.concatenate([np.arange(j, n, dtype=np.int64) for j in range(n)])
vx = rng.standard_normal(ri.size) * 0.01
vx[cp[:-1]] = n * 0.05 + 1.0
F = Factor(n, cp, ri, "L", np.arange(n, dtype=np.int64), {"ordering": 1})
info = F.info()
dev = torch.device('cuda', 0)
v_d = torch.from_numpy(vx).to(dev)
F.factorize_dev(v_d.data_ptr()); F.factorize_dev(v_d.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    F.factorize_dev(v_d.data_ptr())
tf = (time.perf_counter() - t0) / 3
b = rng.standard_normal(n); x_d = torch.from_numpy(b.copy()).to(dev)
F.solve_dev(x_d.data_ptr(), 0, 1, n)
x = x_d.cpu().numpy()
A = np.zeros((n, n)); 
for j in range(n): A[j:, j] = vx[cp[j]:cp[j+1]]
A = A + np.tril(A, -1).T
print(json.dumps({"n": n, "nsuper": int(info["nsuper"]), "max_front": int(info["max_front"]), "factor_ms": tf * 1e3,
                  "TF_s": n ** 3 / 3 / tf / 1e12, "rel_residual": float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))}))
