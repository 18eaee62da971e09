import sys
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer
_lib.require_device()
g, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 9)
n, cp, ri, v = workloads.laplacian_2d(g, h)
F = Factor(n, cp, ri)
F.factorize(v)
info = F.info()
print({k: info[k] for k in ("nsuper", "nlevels", "max_front")})
rng = np.random.default_rng(1)
nr = 64
B = rng.standard_normal((n, nr))
for sysc in (4, 5, 0):
    d = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
    F.solve_dev(d.ptr, sys=sysc, nrhs=nr, ldB=n)
    X = d.download(np.float64, n * nr).reshape((n, nr), order="F")
    Xr = np.empty_like(X)
    for j in range(nr):
        dj = DeviceBuffer.from_array(np.ascontiguousarray(B[:, j]))
        F.solve_dev(dj.ptr, sys=sysc, nrhs=1, ldB=n)
        Xr[:, j] = dj.download(np.float64, n)
    E = np.abs(X - Xr) / np.abs(Xr).max()
    badrows = np.flatnonzero(E.max(axis=1) > 1e-10)
    badcols = np.flatnonzero(E.max(axis=0) > 1e-10)
    print("sys", sysc, "max err %.3e" % E.max(), "bad rows", len(badrows), "of", n, "first", badrows[:20], "bad cols", len(badcols), badcols[:20])
    if len(badrows):
        r = badrows[0]
        print("  row", r, "wide", X[r, :4], "ref", Xr[r, :4], "B", B[r, :4])
