"""k_atda on the random-pattern calibration of BASELINE.md section 2: G 200 000 x 50 000, 4 entries per row (nnz(S) ~ 1.25e6)."""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from kvxopt_amd import _lib
ml, n = 200000, 50000
rng = np.random.default_rng(4)
rows = np.repeat(np.arange(ml), 4)
cols = rng.integers(0, n, size=4 * ml)
G = sp.csc_matrix((rng.standard_normal(4 * ml), (rows, cols)), shape=(ml, n)); G.sum_duplicates(); G.sort_indices()
Gp, Gi, Gx = G.indptr.astype(np.int64), G.indices.astype(np.int64), G.data.copy()
L = _lib.lib()
h = ctypes.c_void_p()
t0 = time.perf_counter()
_lib.raise_for(L.kvx_atda_plan(ml, n, _lib.pi(Gp), _lib.pi(Gi), None, None, ctypes.byref(h)))
t_plan = time.perf_counter() - t0
snz = ctypes.c_int64()
_lib.raise_for(L.kvx_atda_pattern(h, ctypes.byref(snz), None, None))
gx = _lib.DeviceBuffer.from_array(Gx); w_h = rng.uniform(0.5, 1.5, ml); w = _lib.DeviceBuffer.from_array(w_h)
sx = _lib.DeviceBuffer(8 * snz.value)
for _ in range(5):
    _lib.raise_for(L.kvx_atda_assemble_dev(h, gx.ptr, w.ptr, None, sx.ptr))
L.kvx_dev_sync()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    _lib.raise_for(L.kvx_atda_assemble_dev(h, gx.ptr, w.ptr, None, sx.ptr))
L.kvx_dev_sync()
dt = (time.perf_counter() - t0) / reps
alg = 12.0 * len(Gx) + 8.0 * ml + 8.0 * snz.value
S = sx.download(np.float64, snz.value)
Sref = sp.tril((G.T @ sp.diags(w_h) @ G).tocsc()).tocsc(); Sref.sort_indices()
print("nnz(G) %d nnz(S) %d plan %.3f s; %.2f us per launch (back to back), algorithmic %.1f MB -> %.0f GB/s = %.1f %% of 8 TB/s; max rel err %.1e"
      % (len(Gx), snz.value, t_plan, dt * 1e6, alg / 1e6, alg / dt / 1e9, alg / dt / 8e12 * 100,
         np.abs(S - Sref.data).max() / np.abs(Sref.data).max() if Sref.nnz == snz.value else -1))
