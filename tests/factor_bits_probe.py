"""Helper of test_chol_gpu.py::test_fused_first_diagonal_block_is_bitwise: digests of the factor and of a solve for a few systems
(run once with KVX_ASM_POTRF_WGS=0 -- the separate k_potrf_blk launch -- and once with the default)."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import workloads                        # noqa: E402
from kvxopt_amd.chol import Factor                      # noqa: E402

res = {}
for name, (n, cp, ri, v) in (("lap2d_150", workloads.laplacian_2d(150)), ("lap2d_97x211", workloads.laplacian_2d(97, 211)),
                             ("lap3d_16", workloads.laplacian_3d(16)), ("stencil21_60", workloads.stencil21_2d(60))):
    F = Factor(n, cp, ri)
    F.factorize(v)
    Lp, Li, Lx = F.get_factor()
    b = np.random.default_rng(3).standard_normal(n)
    x = b.copy(); F.solve(x)
    inf = F.info()
    res[name] = {"L": hashlib.sha256(Lx.tobytes()).hexdigest(), "x": hashlib.sha256(x.tobytes()).hexdigest(), "max_front": int(inf["max_front"])}
print(json.dumps(res))
