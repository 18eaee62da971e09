import sys
from lu_proto2 import run
for nm in ("ACTIVSg2000.mtx", "bcsstk13.mtx", "bcsstk24.mtx"):
    for stol in (1e-3, 1e-6):
        print("stol", stol); run(nm, stol=stol, btf=False)
