#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel trace csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_scatter_a' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for r in rows[s:e]:
    nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('kvx::', '')
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    if st < lo or st > hi: continue
    print("%9.1f %8.1f  q%-3s %-22s wg=%d x%s" % (st, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], nm[:22],
          int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y']))
