import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_lu_rowmax' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
t0=int(rows[a]['Start_Timestamp'])
import re, collections
agg=collections.OrderedDict()
for r in rows[a:b]:
    nm=re.sub(r'.*::','',r['Kernel_Name'].split('(')[0])
    if 'k_lu_front' in r['Kernel_Name']: nm='k_lu_front'
    if 'k_lu_fwd' in r['Kernel_Name']: nm='k_lu_fwd'
    if 'k_lu_bwd' in r['Kernel_Name']: nm='k_lu_bwd'
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if len(sys.argv)>2: print("%8.1f %8.1f  %-16s grid %s wg %s"%((int(r['Start_Timestamp'])-t0)/1e3,dur,nm,str(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z'],r['Workgroup_Size_X']))
    c=agg.setdefault(nm,[0,0.0]); c[0]+=1; c[1]+=dur
for k,v in agg.items(): print("%-18s %4d launches %9.1f us"%(k,v[0],v[1]))
print("span us", (int(rows[b]['Start_Timestamp'])-t0)/1e3)
