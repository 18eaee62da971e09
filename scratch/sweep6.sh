#!/bin/bash
mkdir -p gpurun_out/r2m
for z3 in 0.05 0.06 0.07 0.075 0.08 0.09; do
  export KVX_RELAX_Z3=$z3
  for g in 800 1000 1300 1600; do timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --grid $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('z3=$z3 lap2d $g step %.3f factor %.3f solve %.3f levels %d lsize %.3e'%(d['ms_per_step'],d['ms_factor'],d['ms_solve'],d['nlevels'],d['lsize']))"; done
done 2>&1 | tee gpurun_out/r2m/sweep6.log
