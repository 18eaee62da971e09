"""Multi-rhs solve timing: ms per solve for nrhs in a list, on the 250x200 grid (n = 50k) and optionally 1000x1000."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for

def run(g, h, rhs_list, reps=5):
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    F = Factor(n, cp, ri)
    F.factorize(v)
    rng = np.random.default_rng(0)
    for nr in rhs_list:
        B = rng.standard_normal((n, nr))
        d = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
        for _ in range(3):
            F.solve_dev(d.ptr, sys=0, nrhs=nr, ldB=n)
        raise_for(lib().kvx_dev_sync())
        t = time.perf_counter()
        for _ in range(reps):
            F.solve_dev(d.ptr, sys=0, nrhs=nr, ldB=n)
        raise_for(lib().kvx_dev_sync())
        ms = (time.perf_counter() - t) / reps * 1e3
        # check one solve
        d2 = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
        F.solve_dev(d2.ptr, sys=0, nrhs=nr, ldB=n)
        X = d2.download(np.float64, n * nr).reshape((n, nr), order="F")
        R = workloads.sym_matvec(n, cp, ri, v, X) - B
        print("grid %dx%d n=%d nrhs=%d: %.3f ms/solve (%.3f ms per rhs), residual %.2e" % (g, h, n, nr, ms, ms / nr, np.abs(R).max() / np.abs(B).max()), flush=True)

if __name__ == "__main__":
    _lib.require_device()
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        run(250, 200, [1, 2, 4, 8, 32, 200])
    elif which == "one":
        run(1000, 1000, [1], reps=2)
    elif which == "bigprof":
        run(1000, 1000, [64], reps=2)
    elif which == "prof":
        run(250, 200, [200], reps=3)
    else:
        run(1000, 1000, [1, 4, 16, 64], reps=3)
