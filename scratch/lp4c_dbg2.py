import sys, os
sys.path.insert(0, '.')
if os.environ.get("WITH_TORCH") == "1":
    import torch
    print("torch cuda", torch.cuda.is_available(), flush=True)
import numpy as np
from kvxopt_amd import lp, workloads
from kvxopt_amd.base import spmatrix
L = workloads.lp_grid_eq(250, 200, 200)
G = spmatrix.from_ccs(L["ml"], L["n"], L["Gp"], L["Gi"], L["Gx"])
A = spmatrix.from_ccs(L["p"], L["n"], L["Ap"], L["Ai"], L["Ax"])
sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"], options={"maxiters": 2})
print("first ->", sol["status"], sol["iterations"], flush=True)
sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
print("second ->", sol["status"], sol["iterations"], flush=True)
sol = lp.conelp(L["c"], G, L["h"], A=A, b=L["b"])
print("third ->", sol["status"], sol["iterations"], flush=True)
