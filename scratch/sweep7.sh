#!/bin/bash
mkdir -p gpurun_out/r2m
for o in '{}' '{"relax_z2":0.15}' '{"relax_z2":0.2}' '{"relax_z2":0.3}' '{"relax_z1":0.9}' '{"relax_z1":0.6}' '{"relax_small":8}' '{"relax_small":2}' '{"leaf_cols":32,"leaf_rows":96}' '{"leaf_cols":40,"leaf_rows":80}' '{"nd_leaf":128}' '{"nd_leaf":64}' '{"nd_leaf":200}'; do
  for g in 1000 1300; do timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --grid $g --chol-opts "$o" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-36s g=$g step %.3f factor %.3f solve %.3f levels %d lsize %.3e'%(d['opts'],d['ms_per_step'],d['ms_factor'],d['ms_solve'],d['nlevels'],d['lsize']))"; done
done 2>&1 | tee gpurun_out/r2m/sweep7.log
