"""Wall time of the first calls of factorize / solve on one handle: call 1 eager, call 2 captures + instantiates the graph, call 3+ replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("WITH_TORCH") == "1":
    import torch; torch.cuda.is_available()
import numpy as np
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
for g in ((250, 200), (1000, 1000)):
    n, cp, ri, v = workloads.laplacian_2d(*g)
    F = Factor(n, cp, ri)
    vd = _lib.DeviceBuffer.from_array(v); bd = _lib.DeviceBuffer.from_array(np.ones(n))
    ts = []
    for it in range(5):
        t0 = time.perf_counter(); F.factorize_dev(vd.ptr); t1 = time.perf_counter(); F.solve_dev(bd.ptr, 0, 1, n); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print("grid", g, "factor/solve ms per call:", ts, flush=True)
