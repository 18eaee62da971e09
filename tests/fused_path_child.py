"""Child process of test_chol_gpu.py::test_two_enqueue_form_reports_like_the_one_enqueue_form: kvx_chol_factorize_solve on a
definite and on an indefinite matrix, printing which form the call took (kvx_chol_last_fused_path) and what it raised.
WITH_TORCH=1 imports torch first (the process then runs on the HIP runtime of the wheel, where the library must take the
two-enqueue form); KVX_NO_GRAPH=1 forces that form on any runtime."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if os.environ.get("WITH_TORCH") == "1":
    import torch  # noqa: F401
import numpy as np
from kvxopt_amd import _lib, workloads, cholmod
from kvxopt_amd.base import matrix, spmatrix
from kvxopt_amd.chol import Factor
_lib.require_device()
n, cp, ri, v = workloads.laplacian_2d(37, 29)
F = Factor(n, cp, ri)
rng = np.random.default_rng(5)
B = np.asfortranarray(rng.standard_normal((n, 2)))
ref = B.copy(order="F")
F.factorize(v); F.solve(ref)
out = {"paths": [], "equal": True}
for rep in range(3):
    X = B.copy(order="F")
    F.factorize_solve(v, X, nrhs=2, ldB=n)
    out["paths"].append(F.last_fused_path())
    out["equal"] = out["equal"] and bool(np.array_equal(X, ref))
bad = v.copy(); bad[cp[int(F.perm()[n // 3])]] = -2.0
try:
    Factor(n, cp, ri).factorize(bad)
except ArithmeticError as e:
    out["minor_numeric"] = int(e.args[0])
try:
    F.factorize_solve(bad, B.copy(order="F"), nrhs=2, ldB=n)
    out["minor_fused"] = None
except ArithmeticError as e:
    out["minor_fused"] = e.args[0] if isinstance(e.args[0], str) else int(e.args[0])
A = spmatrix.from_ccs(n, n, cp, ri, bad)
try:
    cholmod.linsolve(A, matrix(B.copy(order="F")))
    out["minor_linsolve"] = None
except ArithmeticError as e:
    out["minor_linsolve"] = e.args[0] if isinstance(e.args[0], str) else int(e.args[0])
import ctypes
v_rt = ctypes.c_int(0)
print("RESULT " + json.dumps(out), flush=True)
