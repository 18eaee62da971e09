"""Where the time of an LDS-resident LU front goes: phase counters of k_lu_front_wp.

Needs a development build of the kernels with the counters compiled in:
    cd kvxopt_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DKVX_LU_PHASE -c lu_kernels.hip -o lu_kernels.o && make
(the committed library has none of it).  Run on the GPU box: python3 tools/lu_phase.py
"""
import sys, os, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kvxopt_amd import klu, _lib
from kvxopt_amd.base import spmatrix
z = np.load(os.path.join(ROOT, "tests", "golden", "ACTIVSg2000.npz")); n = int(z["n"])
A = spmatrix.from_ccs(n, n, z["colptr"], z["rowind"], z["values"])
Fs = klu.symbolic(A); Fn = klu.numeric(A, Fs)
vals_d = _lib.DeviceBuffer.from_array(A.values)
lib = ctypes.CDLL(os.path.join(ROOT, "kvxopt_amd", "libkvxhip.so"))
if not hasattr(lib, "kvx_dbg_lu_phase_read"):
    sys.exit("this libkvxhip.so was built without -DKVX_LU_PHASE")
out = (ctypes.c_ulonglong * 16)()
for _ in range(3): Fn.num.refactor_dev(vals_d.ptr, A.values.size)
lib.kvx_dbg_lu_phase_read(out, 1)
N = 10
for _ in range(N): Fn.num.refactor_dev(vals_d.ptr, A.values.size)
lib.kvx_dbg_lu_phase_read(out, 1)
names = {5: "zero + A scatter", 0: "children", 1: "panel (wave 0)", 2: "U12 solve", 3: "rank update", 4: "store"}
tot = sum(out[i] for i in names)
for i, nm in names.items():
    print("%-18s %8.1f us summed over the fronts of a refactorisation  (%4.1f %%)" % (nm, out[i] / 100.0 / N, 100.0 * out[i] / tot))
