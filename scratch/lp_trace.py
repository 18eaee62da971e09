import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from kvxopt_amd import lp, workloads
from kvxopt_amd.base import spmatrix
P = workloads.lp_grid(250, 200)
G = spmatrix.from_ccs(P["ml"], P["n"], P["Gp"], P["Gi"], P["Gx"])
lp.conelp(P["c"], G, P["h"], options={"maxiters": 2})
sol = lp.conelp(P["c"], G, P["h"])
print(sol["iterations"], sol["loop seconds"])
