"""Child process of test_chol_gpu.py::test_graph_replay_with_the_hip_runtime_of_the_torch_wheel: two factors (a sparse one and a small
dense one, as misc.kkt_chol2 keeps for S and K) refactorised alternately through their captured launch graphs, every result checked.
WITH_TORCH=1 imports torch first, so that the process runs on the HIP runtime bundled with the wheel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if os.environ.get("WITH_TORCH") == "1":
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.version.hip, flush=True)
import numpy as np
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for
_lib.require_device()
n, cp, ri, v = workloads.laplacian_2d(250, 200)
FS = Factor(n, cp, ri)
dS = DeviceBuffer.from_array(np.ascontiguousarray(v))
p = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
M = rng.standard_normal((p, p)); K = M @ M.T + p * np.eye(p)
kcp = np.arange(0, p * p + 1, p, dtype=np.int64)[: p + 1]
# full lower pattern, column-major
rows = np.concatenate([np.arange(j, p) for j in range(p)]); kcp = np.concatenate([[0], np.cumsum([p - j for j in range(p)])]).astype(np.int64)
kv = np.concatenate([K[j:, j] for j in range(p)])
FK = Factor(p, kcp, rows.astype(np.int64))
dK = DeviceBuffer.from_array(np.ascontiguousarray(kv))
b = rng.standard_normal(p)
ref = np.linalg.solve(K, b)
kv_bad = kv.copy()
kv_bad[0] = -1.0                                   # K[0, 0] < 0: not positive definite
dKbad = DeviceBuffer.from_array(np.ascontiguousarray(kv_bad))
bad = 0
for it in range(16):
    if it % 5 == 4:                                # a replay that must FAIL, and be reported: the status word travels through the graph too
        FK.factorize_dev(dKbad.ptr, sync=False)
        raise_for(lib().kvx_dev_sync())
        try:
            FK.status()
            bad += 1
            print("iteration", it, "an indefinite matrix went unnoticed", flush=True)
        except ArithmeticError:
            pass
    FS.factorize_dev(dS.ptr, sync=False)
    B = DeviceBuffer.from_array(np.asfortranarray(rng.standard_normal((n, 8))).reshape(-1, order="F"))
    FS.solve_dev(B.ptr, sys=0, nrhs=8, ldB=n, sync=False)
    FK.factorize_dev(dK.ptr, sync=False)
    db = DeviceBuffer.from_array(b.copy())
    FK.solve_dev(db.ptr, sys=0, nrhs=1, ldB=p, sync=False)
    raise_for(lib().kvx_dev_sync())
    try:
        FS.status(); FK.status()
        x = db.download(np.float64, p)
        err = np.abs(x - ref).max() / np.abs(ref).max()
        ok = err < 1e-10
    except ArithmeticError as e:
        ok = False; err = str(e)
    if not ok:
        bad += 1
        print("iteration", it, "FAILED", err, flush=True)
    junk = [DeviceBuffer.from_array(rng.standard_normal(1000 + 100 * it)) for _ in range(5)]      # churn the pool like an IPM call
print("done, failures:", bad)
