#!/usr/bin/env python3
"""Timeline of one interior-point iteration (k_init_factor .. the next k_clear_factor) near the end of a rocprofv3 kernel trace."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "").replace("(anonymous namespace)::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", "0"),
               int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)))
ks.sort()
ci = [i for i, k in enumerate(ks) if k[2].startswith(("k_clear_factor", "k_init_factor"))]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s, e = ci[-back - 1], ci[-back]
t0 = ks[s][0]
print("iteration span %.3f ms, %d kernels" % ((ks[e][0] - t0) / 1e6, e - s))
qmap = {}
prev_end = t0
for st, en, nm, q, gx, gy in ks[s:e]:
    qi = qmap.setdefault(q, len(qmap))
    print("%9.1f %8.1f  gap %6.1f q%-2d %-34s wg=%dx%d" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, qi, nm[:34], gx, gy))
    prev_end = max(prev_end, en)
