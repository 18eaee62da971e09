"""Converts the SuiteSparse-collection DATA files the reference's tests hold
(/root/reference/tests/*.mtx) into .npz CCS fixtures, read exactly as
tests/test_sparse_solvers.py:36-68 does (the `symmetric` header is ignored, so bcsstk13 /
bcsstk24 are lower-triangular general matrices; ACTIVSg2000 and bp_800 are the unsymmetric KLU cases).
Data only; run in the build container."""
import os
import sys

import numpy as np

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))


def read(fn):
    size = None
    I, J, V = [], [], []
    with open(fn) as fd:
        for row in fd:
            if row.startswith("%"):
                continue
            if size is None:
                size = list(map(int, row.split()))
                continue
            a = row.split()
            I.append(int(a[0]) - 1)
            J.append(int(a[1]) - 1)
            V.append(float(a[2]))
    return size, np.array(I), np.array(J), np.array(V)


for name in (sys.argv[1:] or ("bcsstk13", "bcsstk24", "ACTIVSg2000", "bp_800")):
    size, I, J, V = read(os.path.join(REF, name + ".mtx"))
    n = size[0]
    order = np.lexsort((I, J))
    I, J, V = I[order], J[order], V[order]
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, J + 1, 1)
    np.cumsum(colptr, out=colptr)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, colptr=colptr, rowind=I.astype(np.int64), values=V)
    print(name, n, len(V))
