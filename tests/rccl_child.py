"""Child process of tests/test_rccl_gpu.py: a rank of the sharded factor on librccl.so bound directly (kvxopt_amd.rccl.World,
csrc/rccl_comm.cpp).  One GPU per box, and RCCL refuses two ranks on one device: the rehearsal is a world of ONE rank -- every
call still goes through ncclCommInitRank / ncclBroadcast / ncclAllReduce / ncclAllGather of the real library, from C, on the
system HIP runtime.  Prints RESULT {...}."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from kvxopt_amd import _lib, workloads
from kvxopt_amd.rccl import World
from kvxopt_amd.dist import DistFactor
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, DistOp, lib, raise_for

out = {}
W = World()                                          # RANK / WORLD_SIZE from the environment (1 rank)
out["world"], out["rank"], out["version"] = W.world, W.rank, W.version
W.barrier()
out["max"] = W.max(3.5 + W.rank)
out["gather"] = W.all_gather([1.0 + W.rank, 2.0]).tolist()
# the callback itself, as the library calls it: broadcast and all-reduce on a device buffer
buf = DeviceBuffer.from_array(np.arange(8, dtype=np.float64))
for kind in (1, 2, 3):
    op = DistOp(kind, 0, 0, W.world, 8, buf.ptr)
    raise_for(lib().kvx_rccl_comm(W.comm_ctx, ctypes.byref(op)), "kvx_rccl_comm")
raise_for(lib().kvx_dev_sync())
out["buf"] = buf.download(np.float64, 8).tolist()
out["direct_calls"] = W.stats()[0]
# the sharded factor over this world: same answer as the single-GPU factor
n, cp, ri, v = workloads.laplacian_2d(60, 45)
DF = DistFactor(n, cp, ri, comm=W)
vd = DeviceBuffer.from_array(v)
b = np.random.default_rng(1).standard_normal(n)
xd = DeviceBuffer.from_array(b)
DF.factorize(vd)
DF.solve(xd)
x = xd.download(np.float64, n)
F = Factor(n, cp, ri); F.factorize(v)
xr = b.copy(); F.solve(xr)
out["solve_equal"] = bool(np.array_equal(x, xr))
out["residual"] = float(np.linalg.norm(workloads.sym_matvec(n, cp, ri, v, x.reshape(n, 1)).ravel() - b) / np.linalg.norm(b))
bad = v.copy(); bad[cp[int(F.perm()[n // 2])]] = -1.0
try:
    Factor(n, cp, ri).factorize(bad)
except ArithmeticError as e:
    out["minor_single"] = int(e.args[0])
vd.upload(bad)
try:
    DF.factorize(vd)
    out["minor_dist"] = None
except ArithmeticError as e:
    out["minor_dist"] = int(e.args[0])
out["collectives"] = DF.collectives
out["torch_loaded"] = "torch" in sys.modules
out["backend"] = DF.backend
W.close()
print("RESULT " + json.dumps(out), flush=True)
