"""One host analysis with per-phase wall times (KVX_ANALYZE_TIMING=1): python scratch/analyze_once.py lap3d200"""
import sys, time, os
sys.path.insert(0, ".")
from kvxopt_amd import workloads
from kvxopt_amd.chol import Factor
which = sys.argv[1]
t = time.time()
n, cp, ri, v = workloads.laplacian_2d(1000) if which == "lap2d" else workloads.laplacian_3d(int(which[5:]))
print("generated in %.2f s" % (time.time() - t), flush=True)
t = time.time(); F = Factor(n, cp, ri); print("%s analysis total %.3f s" % (which, time.time() - t), flush=True)
print({k: F.info()[k] for k in ("nsuper", "nlevels", "lnz", "max_front")})
