"""Cycle budget of the workgroup that carries the pivot chain (tile (0, 0) of k_syrk_trailing: update, then the next diagonal block):
run with the library built by scratch/phase_timing.sh."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["KVX_LIB_PATH"] = os.path.join(HERE, "libkvxhip_phase.so")
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from kvxopt_amd import _lib, workloads
from kvxopt_amd.chol import Factor
from kvxopt_amd._lib import DeviceBuffer, lib, raise_for
_lib.require_device()
L = ctypes.CDLL(os.environ["KVX_LIB_PATH"])
L.kvx_dbg_phase_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
n, cp, ri, v = workloads.laplacian_2d(1000)
F = Factor(n, cp, ri)
dv = DeviceBuffer.from_array(np.ascontiguousarray(v))
for _ in range(3):
    F.factorize_dev(dv.ptr, sync=True)
out = (ctypes.c_ulonglong * 16)()
L.kvx_dbg_phase_read(out, 1)
steps = 10
for _ in range(steps):
    F.factorize_dev(dv.ptr, sync=True)
L.kvx_dbg_phase_read(out, 0)
cnt = out[0]
ghz = 2.4
names = {1: "descriptor + operands + MFMA update", 2: "read-modify-write of the tile", 3: "tile into LDS", 4: "potrf_lds (whole)", 5: "store factor + inverse",
         6: "  phase A: pivot sweeps", 7: "  phase B: tiles below + inverse row", 8: "  phase C: trailing tiles"}
print("fused workgroups per factorisation: %.1f" % (cnt / steps))
tot = sum(out[i] for i in (1, 2, 3, 4, 5))
for i in (1, 2, 3, 4, 5, 6, 7, 8):
    print("%-40s %8.0f cycles = %6.2f us per fused workgroup" % (names[i], out[i] / cnt, out[i] / cnt / ghz / 1e3))
print("%-40s %8.0f cycles = %6.2f us" % ("sum of [1..5]", tot / cnt, tot / cnt / ghz / 1e3))
for i, nm in ((9, "  [1] descriptor + set-up"), (10, "  [1] k-steps 0-3: loads + MFMAs"), (11, "  [1] k-steps 4-7"), (12, "  [1] k-steps 8-11"), (13, "  [1] k-steps 12-15")):
    print("%-40s %8.0f cycles = %6.2f us" % (nm, out[i] / cnt, out[i] / cnt / ghz / 1e3))
