import sys, numpy as np
sys.path.insert(0, "/root/repo")
from kvxopt_amd import klu, _lib, workloads
from kvxopt_amd.base import spmatrix
n, cp, ri, v = workloads.convdiff_2d(600)
A = spmatrix.from_ccs(n, n, cp, ri, v)
Fs = klu.symbolic(A); Fn = klu.numeric(A, Fs)
vals_d = _lib.DeviceBuffer.from_array(v)
b_d = _lib.DeviceBuffer.from_array(np.random.default_rng(1).standard_normal(n))
for _ in range(3):
    Fn.num.refactor_dev(vals_d.ptr, v.size)
    Fn.num.solve_dev(b_d.ptr, "N", 1)
