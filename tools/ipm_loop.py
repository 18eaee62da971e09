#!/usr/bin/env python3
"""The device-resident conelp of the bench's IPM leg (grid LP 250x200) a few times: for rocprofv3 traces (tools/trace_ipm.sh)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import lp as kvx_lp, workloads
from kvxopt_amd.base import spmatrix
Pl = workloads.lp_grid(250, 200)
Gl = spmatrix.from_ccs(Pl["ml"], Pl["n"], Pl["Gp"], Pl["Gi"], Pl["Gx"])
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    t0 = time.perf_counter()
    s = kvx_lp.conelp(Pl["c"], Gl, Pl["h"])
    print("call %d: %d iterations, loop %.1f it/s, whole call %.1f it/s" % (rep, s["iterations"], s["iterations"] / s["loop seconds"],
                                                                          s["iterations"] / (time.perf_counter() - t0)),
          "phases us/it:", [round(1e6 * t / s["iterations"], 1) for t in s["phase seconds"]], flush=True)
