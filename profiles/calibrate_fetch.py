#!/usr/bin/env python3
"""Known-byte-count kernels for the FETCH_SIZE / WRITE_SIZE calibration MI355X_MICROARCH.md asks for ("other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern"): the library's loads are 8 bytes per lane (512 B per
wavefront instruction).  k_scal reads and writes N doubles in exactly that pattern; N = 2^26 (512 MiB each way, twice the
Infinity Cache) five times.  profiles/summarize.py divides the known bytes by the counter values of these launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kvxopt_amd import _lib
L = _lib.lib()
_lib.require_device()
N = 1 << 26
x = _lib.DeviceBuffer(8 * N)
_lib.raise_for(L.kvx_vec_fill_dev(N, 1.0, x.ptr))
for _ in range(5):
    _lib.raise_for(L.kvx_vec_scal_dev(N, 1.0000001, x.ptr))
_lib.raise_for(L.kvx_dev_sync())
print("calibration: k_scal on", N, "doubles, 5 launches;", 8 * N, "bytes read and", 8 * N, "bytes written per launch")
