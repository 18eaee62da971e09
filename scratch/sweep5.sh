#!/bin/bash
# robustness of the relaxed-amalgamation bound z3 across workloads
mkdir -p gpurun_out/r2m
for z3 in 0.05 0.075 0.1; do
  export KVX_RELAX_Z3=$z3
  echo "== z3=$z3"
  for g in 700 1000 1300; do timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --grid $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lap2d $g step %.3f factor %.3f solve %.3f levels %d lsize %.3e'%(d['ms_per_step'],d['ms_factor'],d['ms_solve'],d['nlevels'],d['lsize']))"; done
  timeout -k 5 200 python bench.py --quick --steps 6 --warmup 2 --workload lap3d --grid 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lap3d 60 step %.3f factor %.3f solve %.3f levels %d lsize %.3e'%(d['ms_per_step'],d['ms_factor'],d['ms_solve'],d['nlevels'],d['lsize']))"
  timeout -k 5 300 python bench_extra.py --cases chol21,lp4b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'ms_per_step' in d: print(d['case'],'step %.3f factor %.3f solve %.3f levels %d'%(d['ms_per_step'],d['ms_factor'],d['ms_solve'],d['nlevels']))
    else: print(d['case'],'it/s loop %.1f factor %.3f ms solve %.3f ms iters %d'%(d['iterations_per_s_loop_only'],d['ms_kkt_factor'],d['ms_kkt_solve'],d['iterations']))"
done 2>&1 | tee gpurun_out/r2m/sweep5.log
