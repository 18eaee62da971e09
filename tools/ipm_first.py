#!/usr/bin/env python3
"""First call of conelp on a new constraint structure in a warm process (what bench.py reports as ipm.first_call): per-iteration times
on stderr with KVX_LP_TRACE=1."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import lp as kvx_lp, workloads
from kvxopt_amd.base import spmatrix
Pl = workloads.lp_grid(250, 200)
Gl = spmatrix.from_ccs(Pl["ml"], Pl["n"], Pl["Gp"], Pl["Gi"], Pl["Gx"])
kvx_lp.conelp(Pl["c"], Gl, Pl["h"], options={"maxiters": 2})
kvx_lp.clear_cache()
print("---- first call on a new structure", file=sys.stderr)
t0 = time.perf_counter()
s = kvx_lp.conelp(Pl["c"], Gl, Pl["h"])
print("first call: %d iterations, loop %.1f it/s (%.1f ms), whole call %.1f it/s (%.1f ms)" % (
    s["iterations"], s["iterations"] / s["loop seconds"], 1e3 * s["loop seconds"], s["iterations"] / (time.perf_counter() - t0),
    1e3 * (time.perf_counter() - t0)), flush=True)
