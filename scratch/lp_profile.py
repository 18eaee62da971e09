import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np
from kvxopt_amd import lp, workloads
from kvxopt_amd.base import spmatrix
P = workloads.lp_grid(250, 200)
ml, n = P["ml"], P["n"]
cols = np.repeat(np.arange(n), np.diff(P["Gp"]))
G = spmatrix(P["Gx"], P["Gi"], cols, (ml, n))
lp.conelp(P["c"], G, P["h"], options={"maxiters": 2})
t0 = time.perf_counter(); sol = lp.conelp(P["c"], G, P["h"]); dt = time.perf_counter() - t0
print("iterations", sol["iterations"], "wall", dt)
pr = cProfile.Profile(); pr.enable(); sol = lp.conelp(P["c"], G, P["h"]); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
