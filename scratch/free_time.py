import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from kvxopt_amd import workloads, _lib
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import lib
_lib.require_device()
n, cp, ri, v = workloads.laplacian_2d(250, 200)
for rep in range(3):
    t = time.perf_counter(); F = Factor(n, cp, ri); t_an = time.perf_counter() - t
    t = time.perf_counter(); F.factorize(v); t_f1 = time.perf_counter() - t
    t = time.perf_counter(); F.factorize(v); t_f2 = time.perf_counter() - t
    t = time.perf_counter(); F.factorize(v); t_f3 = time.perf_counter() - t
    x = np.ones(n)
    t = time.perf_counter(); F.solve(x); t_s1 = time.perf_counter() - t
    t = time.perf_counter(); F.solve(x); t_s2 = time.perf_counter() - t
    t = time.perf_counter(); F.solve(x); t_s3 = time.perf_counter() - t
    t = time.perf_counter(); lib().kvx_chol_free(F._h); F._h = None; t_free = time.perf_counter() - t
    print("analysis %.1f | factor %.1f %.1f %.1f | solve %.1f %.1f %.1f | free %.1f ms" % tuple(1e3 * a for a in (t_an, t_f1, t_f2, t_f3, t_s1, t_s2, t_s3, t_free)), flush=True)
