import sys, os
sys.path.insert(0, '.')
import numpy as np
from kvxopt_amd import lp, workloads
from kvxopt_amd.base import spmatrix
L = workloads.lp_grid_eq(250, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 200)
G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
for mi in (2, 3, 5, 100, 100):
    sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"], options={"maxiters": mi, "show_progress": False})
    print("maxiters", mi, "->", sol["status"], sol["iterations"], "gap %.3e" % (sol["gap"] or -1), flush=True)
