import sys
sys.path.insert(0, "/root/repo")
from kvxopt_amd import mps
st = {}
for i in range(12):
    out = mps.solve("/root/repo/tests/golden/boeing2.mps")
    st.setdefault((out["status"], out["iterations"]), []).append(round(out.get("objective", 0), 6))
print(st)
