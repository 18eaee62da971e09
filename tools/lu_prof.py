import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from kvxopt_amd import klu, _lib
from kvxopt_amd.base import spmatrix
z = np.load("/root/repo/tests/golden/ACTIVSg2000.npz"); n = int(z["n"])
A = spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])
Fs = klu.symbolic(A); Fn = klu.numeric(A, Fs)
vals_d = _lib.DeviceBuffer.from_array(A.values)
B = np.asfortranarray(np.random.default_rng(3).standard_normal((n, 3)))
b_d = _lib.DeviceBuffer.from_array(B.reshape(-1, order="F"))
for _ in range(10):
    Fn.num.refactor_dev(vals_d.ptr, A.values.size)
    Fn.num.solve_dev(b_d.ptr, "N", 3)
P = Fn.num.sym  # keep alive
import collections
