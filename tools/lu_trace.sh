#!/bin/bash
# per-launch durations of one KLU refactorisation + solve (ACTIVSg2000), in launch order, beside the plan's levels
# (KVX_LU_DUMP_PLAN=1).  Run through gpurun from the repo root: bash tools/lu_trace.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KVX_LU_DUMP_PLAN=1 python3 tools/lu_prof.py 2>&1 | grep "lu level" | head -12
rm -rf gpurun_out/lutr
rocprofv3 --kernel-trace -d gpurun_out/lutr -o t --output-format csv -- python3 tools/lu_prof.py > gpurun_out/lutr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/lutr/**/t_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last refactor: from the last k_lu_rowmax on
idx=[i for i,r in enumerate(rows) if 'k_lu_rowmax' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+40]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print('%9.1f us  +%7.1f us  %s  grid %s' % ((s-t0)/1e3,(e-s)/1e3,r['Kernel_Name'][:60],r.get('Grid_Size_X','?')))
PY
