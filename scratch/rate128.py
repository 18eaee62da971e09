import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
r128=[r for r in rows if 'syrk_trailing128' in r['Kernel_Name']]
r128.sort(key=lambda r:int(r['Start_Timestamp']))
n=int(sys.argv[2])
tot=0;fl=0
for i,r in enumerate(r128[:n//256-1]):
    rowsleft=n-256*(i+1)
    T=(rowsleft+127)//128; tiles=T*(T+1)//2
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=dur; fl+=tiles*2*128*128*256
print("outer updates of one factorisation: %.2f ms, %.1f TF/s"%(tot/1e3, fl/tot/1e6))
