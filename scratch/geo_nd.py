"""How much does the ordering leave on the table?  Config 2 with a geometric nested dissection (straight separators, recursive
coordinate bisection) handed in as the user's permutation, against the library's level-set dissection."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor

def geo_nd(g, h, leaf=8):
    order = []
    def rec(i0, i1, j0, j1):                       # half-open ranges of grid coordinates
        ni, nj = i1 - i0, j1 - j0
        if ni <= 0 or nj <= 0: return
        if ni <= leaf and nj <= leaf:
            for j in range(j0, j1):
                for i in range(i0, i1): order.append(i + g * j)
            return
        if ni >= nj:
            mid = (i0 + i1) // 2
            rec(i0, mid, j0, j1); rec(mid + 1, i1, j0, j1)
            for j in range(j0, j1): order.append(mid + g * j)
        else:
            mid = (j0 + j1) // 2
            rec(i0, i1, j0, mid); rec(i0, i1, mid + 1, j1)
            for i in range(i0, i1): order.append(i + g * mid)
    import sys as _s
    _s.setrecursionlimit(10000)
    rec(0, g, 0, h)
    return np.array(order, dtype=np.int64)

if __name__ == "__main__":
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n, cp, ri, v = workloads.laplacian_2d(g)
    for name, perm in (("library ND", None), ("geometric ND leaf 8", geo_nd(g, g, 8)), ("geometric ND leaf 4", geo_nd(g, g, 4))):
        F = Factor(n, cp, ri, "L", perm)
        i = F.info()
        sup, nrows, parent, level = F.supernodes()
        k = np.diff(sup)
        # chain: sum over levels of the largest pivot count of a big front
        big = (nrows > 128) | (k > 64)
        chain = sum(int(k[(level == l) & big].max()) for l in range(level.max() + 1) if ((level == l) & big).any())
        line = "%-22s lnz %.3e flops %.3e levels %d nsuper %d max_front %d  chain columns %d (%d panels)" % (name, i["lnz"], i["flops"], i["nlevels"], i["nsuper"], i["max_front"], chain, sum(-(-int(k[(level == l) & big].max()) // 64) for l in range(level.max() + 1) if ((level == l) & big).any()))
        try:
            from kvxopt_amd import _lib
            _lib.require_device()
            from kvxopt_amd._lib import DeviceBuffer, lib, raise_for
            dv = DeviceBuffer.from_array(np.ascontiguousarray(v))
            b = np.random.default_rng(0).standard_normal(n)
            db = DeviceBuffer.from_array(b)
            for _ in range(4):
                F.factorize_dev(dv.ptr, sync=False); F.solve_dev(db.ptr, 0, 1, n)
            raise_for(lib().kvx_dev_sync())
            t = time.perf_counter()
            for _ in range(20):
                F.factorize_dev(dv.ptr, sync=False); F.solve_dev(db.ptr, 0, 1, n)
            raise_for(lib().kvx_dev_sync())
            ms = (time.perf_counter() - t) / 20 * 1e3
            f, s = F.timing()
            line += "  step %.3f ms (factor %.3f solve %.3f)" % (ms, f, s)
        except Exception as e:
            line += "  (no device: %s)" % type(e).__name__
        print(line, flush=True)
