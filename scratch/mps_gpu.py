import sys, time, json
sys.path.insert(0, "/root/repo")
from kvxopt_amd import mps
for sp in (False, True, False):
    t0 = time.time()
    out = mps.solve("/root/repo/tests/golden/boeing2.mps", options={"show_progress": sp} if sp else None)
    print("status", out["status"], "iterations", out["iterations"], "objective", out.get("objective"), "time %.3f" % (time.time() - t0), flush=True)
    print({k: out["sol"][k] for k in ("gap", "relative gap", "primal infeasibility", "dual infeasibility", "residual as primal infeasibility certificate")}, flush=True)
