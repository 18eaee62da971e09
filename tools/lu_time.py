#!/usr/bin/env python3
"""Wall time of refactor + solve for the KLU widening cases: 600 x 600 convection-diffusion (BASELINE configs[2] scale) and ACTIVSg2000."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from kvxopt_amd import klu, _lib, workloads
from kvxopt_amd.base import spmatrix

def run(name, n, cp, ri, v):
    A = spmatrix.from_ccs(n, n, cp, ri, v)
    Fs = klu.symbolic(A); Fn = klu.numeric(A, Fs)
    vals_d = _lib.DeviceBuffer.from_array(np.asarray(v, dtype=np.float64))
    b_d = _lib.DeviceBuffer.from_array(np.random.default_rng(1).standard_normal(n))
    for _ in range(3):
        Fn.num.refactor_dev(vals_d.ptr, len(v)); Fn.num.solve_dev(b_d.ptr, "N", 1)
    t0 = time.perf_counter()
    for _ in range(10): Fn.num.refactor_dev(vals_d.ptr, len(v))
    t1 = time.perf_counter()
    for _ in range(10): Fn.num.solve_dev(b_d.ptr, "N", 1)
    t2 = time.perf_counter()
    print("%s: refactor %.2f ms, solve %.3f ms" % (name, (t1 - t0) * 100, (t2 - t1) * 100), flush=True)

run("convdiff 600x600", *workloads.convdiff_2d(600))
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ACTIVSg2000.npz"))
run("ACTIVSg2000", int(z["n"]), z["colptr"], z["rowind"], z["values"])
