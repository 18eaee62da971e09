#!/usr/bin/env python3
"""Timeline of the last bench step (k_init_factor .. k_perm_scatter) from a rocprofv3 kernel trace csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Queue_Id", "0"),
               int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1),
               int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
ks.sort()
ci = [i for i, k in enumerate(ks) if k[2].startswith(("k_clear_factor", "k_init_factor"))]
si = [i for i, k in enumerate(ks) if k[2].startswith("k_perm_scatter")]
s = ci[-1]
e = max(i for i in si if i > s) if any(i > s for i in si) else len(ks) - 1
if e == len(ks) - 1 and len(ci) > 1:
    s = ci[-2]; e = max(i for i in si if s < i < ci[-1])
t0 = ks[s][0]
print("step span %.3f ms, %d kernels" % ((ks[e][1] - t0) / 1e6, e - s + 1))
qmap = {}
for st, en, nm, q, gx, gy, gz, vg, lds in ks[s:e + 1]:
    qi = qmap.setdefault(q, len(qmap))
    print("%9.1f %8.1f  q%-2d %-28s wg=%dx%dx%d v%s l%s" % ((st - t0) / 1e3, (en - st) / 1e3, qi, nm[:28], gx, gy, gz, vg, lds))
