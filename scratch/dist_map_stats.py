"""Host-only: flop shares of the sharded factorisation (kvx_chol_dist_map) for DESIGN.md section 6."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kvxopt_amd import workloads, dist as kd
from kvxopt_amd.chol import Factor

def show(name, F, P, ob, min_m):
    m = kd.partition(F, P, ob, min_m)
    sh = m["rank_flops"] / m["flops"]
    g = m["ghi"] - m["glo"]
    print("%s P=%d ob=%d min_m=%d: max share %.4f (ideal %.4f, speed-up bound %.2fx) min %.4f | replicated %.4f | panel(max rank) %.4f | shared fronts %d, cyclic %d"
          % (name, P, ob, min_m, sh.max(), 1.0 / P, 1.0 / sh.max(), sh.min(), m["replicated"] / m["flops"],
             (m["panel_flops"] / m["flops"]).max(), int((g > 1).sum()), int(m["mode"].sum())))

which = sys.argv[1] if len(sys.argv) > 1 else "lap3d100"
t = time.time()
if which.startswith("lap3d"):
    n, cp, ri, v = workloads.laplacian_3d(int(which[5:]))
else:
    n, cp, ri, v = workloads.laplacian_2d(int(which[5:]))
F = Factor(n, cp, ri)
print(which, "n=%d analysis %.1f s" % (n, time.time() - t), {k: F.info()[k] for k in ("nsuper", "nlevels", "max_front", "flops")})
for P in (2, 4, 8):
    for ob, mm in ((512, 6144), (256, 2048), (1024, 6144)):
        show(which, F, P, ob, mm)
