"""wide solves behind an asynchronous factorisation (what lp.KKTGenEqDev does) + lp4c status with / without the wide path"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for
_lib.require_device()
n, cp, ri, v = workloads.laplacian_2d(250, 200)
F = Factor(n, cp, ri)
dv = DeviceBuffer.from_array(np.ascontiguousarray(v))
rng = np.random.default_rng(0)
nr = 200
B = rng.standard_normal((n, nr))
for rep in range(3):
    d = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
    F.factorize_dev(dv.ptr, sync=False)
    F.solve_dev(d.ptr, sys=0, nrhs=nr, ldB=n, sync=False)
    raise_for(lib().kvx_dev_sync())
    F.status()
    X = d.download(np.float64, n * nr).reshape((n, nr), order="F")
    R = workloads.sym_matvec(n, cp, ri, v, X) - B
    print("rep", rep, "async factor + async 200-rhs solve: residual %.2e" % (np.abs(R).max() / np.abs(B).max()), flush=True)
import bench_extra
for wf in ("0", None):
    if wf is None: os.environ.pop("KVX_WIDE_FROM", None)
    else: os.environ["KVX_WIDE_FROM"] = wf
    from kvxopt_amd import lp
    lp.clear_cache()
    r = bench_extra.lp_eq_case(250, 200, 200)
    print("KVX_WIDE_FROM", wf, r if r is None else {k: r[k] for k in ("status", "iterations", "value")}, flush=True)
