"""narrow vs rhs-major path for 8 .. 48 right-hand sides at several n (KVX_WIDE_FROM is read at the device set-up of a factor)"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for
_lib.require_device()
for g, h in ((250, 200), (500, 500), (1000, 1000)):
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    res = {}
    for wf in ("0", "8"):
        os.environ["KVX_WIDE_FROM"] = wf
        F = Factor(n, cp, ri)
        F.factorize(v)
        rng = np.random.default_rng(0)
        for nr in (8, 16, 32, 48):
            B = rng.standard_normal((n, nr))
            d = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
            for _ in range(3):
                F.solve_dev(d.ptr, sys=0, nrhs=nr, ldB=n)
            raise_for(lib().kvx_dev_sync())
            t = time.perf_counter()
            for _ in range(5):
                F.solve_dev(d.ptr, sys=0, nrhs=nr, ldB=n)
            raise_for(lib().kvx_dev_sync())
            res[(wf, nr)] = (time.perf_counter() - t) / 5 * 1e3
        del F
    for nr in (8, 16, 32, 48):
        print("n=%-8d nrhs=%-3d narrow %.3f ms   rhs-major %.3f ms" % (n, nr, res[("0", nr)], res[("8", nr)]), flush=True)
