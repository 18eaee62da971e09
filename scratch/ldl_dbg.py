import numpy as np, sys
sys.path.insert(0, '.')
from kvxopt_amd import cholmod, workloads
from kvxopt_amd.base import matrix, spmatrix
A = spmatrix([10, 3, 5, -2, 5, 2], [0, 2, 1, 3, 2, 3], [0, 0, 1, 1, 2, 3])
for mode in (2, 0):
    cholmod.options["supernodal"] = mode
    F = cholmod.symbolic(A)
    cholmod.numeric(A, F)
    print("mode", mode, F.fac.info()["is_ll"], flush=True)
    for s in (0, 4, 5, 2, 3, 6):
        X = matrix(1.0, (4, 1))
        print(" sys", s, flush=True)
        cholmod.solve(F, X, sys=s)
        print("   ", X._a, flush=True)
