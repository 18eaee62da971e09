"""Where the set-up time of a first lp.conelp call on a new structure goes (config 4b)."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from kvxopt_amd import lp, workloads, _lib
from kvxopt_amd.base import spmatrix
_lib.require_device()
L = workloads.lp_grid(250, 200)
G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
# warm the process on another structure
Lw = workloads.lp_grid(60, 50)
lp.conelp(Lw["c"], spmatrix.from_ccs(Lw["ml"], Lw["n"], Lw["Gp"], Lw["Gi"], Lw["Gx"]), Lw["h"], options={"maxiters": 3})
for rep in range(2):
    lp.clear_cache()
    t0 = time.perf_counter()
    kkt = lp.KKTChol2Dev(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
    t1 = time.perf_counter()
    del kkt
    lp.clear_cache()
    t2 = time.perf_counter()
    sol = lp.conelp(L["c"], G, L["h"])
    t3 = time.perf_counter()
    print("rep %d: KKT object alone %.1f ms; conelp whole call %.1f ms, loop %.1f ms, %d iterations" % (rep, 1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * sol["loop seconds"], sol["iterations"]), flush=True)
