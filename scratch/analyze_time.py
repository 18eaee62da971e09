"""Host analysis time by phase (KVX_ANALYZE_TIMING=1) on the GPU box's CPU cores."""
import sys, time, os
sys.path.insert(0, ".")
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
which = sys.argv[1]
n, cp, ri, v = workloads.laplacian_2d(1000) if which == "lap2d" else workloads.laplacian_3d(int(which[5:]))
for i in range(3):
    t = time.time(); F = Factor(n, cp, ri); print("%s total %.3f s (ND threads %s, analyze threads %s)" % (which, time.time() - t, os.environ.get("KVX_ND_THREADS", "default"), os.environ.get("KVX_ANALYZE_THREADS", "default")), flush=True)
