"""klu.linsolve on ACTIVSg2000 (BASELINE configs[2]): first call on a new pattern (analysis + factor + solve, host buffers) in a warm
process, and the phases of the host analysis (KVX_ANALYZE_TIMING=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvxopt_amd import klu, workloads
from kvxopt_amd.base import matrix, spmatrix
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ACTIVSg2000.npz"))
n, cp, ri, v = int(g["n"]), g["colptr"], g["rowind"], g["values"]
A = spmatrix.from_ccs(n, n, cp, ri, v)
# warm the process (HIP runtime, code objects, pools) on another matrix
n2, cp2, ri2, v2 = workloads.convdiff_2d(40)
B2 = matrix(np.ones((n2, 1)))
klu.linsolve(spmatrix.from_ccs(n2, n2, cp2, ri2, v2), B2)
rng = np.random.default_rng(3)
for rep in range(3):
    vv = v * (1.0 + 1e-3 * rep)                     # same pattern: calls after the first reuse the cached analysis
    Ar = spmatrix.from_ccs(n, n, cp, ri, vv)
    B = matrix(rng.standard_normal((n, 3)))
    t = time.perf_counter()
    if rep == 0 and os.environ.get('KVX_LU_TIMING'):
        t0 = time.perf_counter(); Fs = klu.symbolic(Ar); t1 = time.perf_counter(); Fn = klu.numeric(Ar, Fs); t2 = time.perf_counter(); klu.solve(Ar, Fs, Fn, B); t3 = time.perf_counter()
        print('  first call pieces: symbolic %.2f numeric %.2f solve %.2f ms' % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)), flush=True)
        del Fs, Fn
        t = time.perf_counter()
    klu.linsolve(Ar, B)
    print("linsolve call %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t)), flush=True)
import scipy.sparse as sp, scipy.sparse.linalg as spla
M = sp.csc_matrix((v, ri, cp), shape=(n, n))
b = rng.standard_normal((n, 3))
t = time.perf_counter(); lu = spla.splu(M); x = lu.solve(b); print("scipy splu + solve: %.2f ms" % (1e3 * (time.perf_counter() - t)))
# phases of a first call, separately (a new pattern again: the 150^2 convection-diffusion grid is not cached yet either)
for name, (nn, c, r, vals) in (("ACTIVSg2000 (values perturbed, cached analysis dropped)", (n, cp, ri, v)),):
    klu._cache.clear() if hasattr(klu, "_cache") else None
    Ar = spmatrix.from_ccs(nn, nn, c, r, vals * 1.01)
    B = matrix(rng.standard_normal((nn, 3)))
    t0 = time.perf_counter(); Fs = klu.symbolic(Ar)
    t1 = time.perf_counter(); Fn = klu.numeric(Ar, Fs)
    t2 = time.perf_counter(); klu.solve(Ar, Fs, Fn, B)
    t3 = time.perf_counter()
    print("%s: symbolic %.2f ms, numeric %.2f ms, solve %.2f ms" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    t2 = time.perf_counter(); klu.numeric(Ar, Fs, Fn); t3 = time.perf_counter()
    print("   refactor %.2f ms" % (1e3 * (t3 - t2)))
