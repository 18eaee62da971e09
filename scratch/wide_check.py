"""Many-rhs (rhs-major, kernels_wide.hip) solves against column-by-column solves: python scratch/wide_check.py [big]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for


def check(name, n, cp, ri, v, nrs=(64, 70, 130), systems=(0, 4, 5)):
    F = Factor(n, cp, ri)
    F.factorize(v)
    rng = np.random.default_rng(1)
    worst = 0.0
    for nr in nrs:
        B = rng.standard_normal((n, nr))
        for sysc in systems:
            d = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
            F.solve_dev(d.ptr, sys=sysc, nrhs=nr, ldB=n)
            X = d.download(np.float64, n * nr).reshape((n, nr), order="F")
            Xr = np.empty_like(X)
            for j in range(nr) if n <= 60000 else (0, nr // 2, nr - 1):
                dj = DeviceBuffer.from_array(np.ascontiguousarray(B[:, j]))
                F.solve_dev(dj.ptr, sys=sysc, nrhs=1, ldB=n)
                Xr[:, j] = dj.download(np.float64, n)
                err = np.abs(X[:, j] - Xr[:, j]).max() / max(np.abs(Xr[:, j]).max(), 1e-300)
                worst = max(worst, err)
                if not err < 1e-10:
                    print("  MISMATCH %s nrhs=%d sys=%d col=%d rel err %.3e" % (name, nr, sysc, j, err), flush=True)
                    bad = np.argmax(np.abs(X[:, j] - Xr[:, j]))
                    print("   first bad row", int(np.flatnonzero(np.abs(X[:, j] - Xr[:, j]) > 1e-8 * np.abs(Xr[:, j]).max())[0]) if err > 1e-8 else -1, "worst row", int(bad))
                    return False
    print("%-28s n=%-8d ok, worst relative difference to single-rhs solves %.2e" % (name, n, worst), flush=True)
    return True


def timing(g, h, nrs, reps=3):
    n, cp, ri, v = workloads.laplacian_2d(g, h)
    F = Factor(n, cp, ri)
    F.factorize(v)
    rng = np.random.default_rng(0)
    for nr in nrs:
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
        d2 = DeviceBuffer.from_array(np.asfortranarray(B).reshape(-1, order="F"))
        F.solve_dev(d2.ptr, sys=0, nrhs=nr, ldB=n)
        X = d2.download(np.float64, n * nr).reshape((n, nr), order="F")
        R = workloads.sym_matvec(n, cp, ri, v, X) - B
        print("grid %dx%d n=%d nrhs=%d: %.3f ms/solve, residual %.2e" % (g, h, n, nr, ms, np.abs(R).max() / np.abs(B).max()), flush=True)


if __name__ == "__main__":
    _lib.require_device()
    ok = True
    ok &= check("grid 12x9", *workloads.laplacian_2d(12, 9))
    ok &= check("grid 60x50", *workloads.laplacian_2d(60, 50))
    ok &= check("grid 250x200", *workloads.laplacian_2d(250, 200), nrs=(64, 130))
    ok &= check("cube 20", *workloads.laplacian_3d(20), nrs=(70,))
    ok &= check("cube 40 (big fronts)", *workloads.laplacian_3d(40), nrs=(64,), systems=(0,))
    if ok and len(sys.argv) > 1:
        timing(250, 200, [64, 200])
        timing(1000, 1000, [64, 256])
    print("ALL OK" if ok else "FAILED")
