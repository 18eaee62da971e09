#!/bin/bash
# analysis-option sweep on the GPU box (round 2: after the packed LDS image)
mkdir -p gpurun_out/r2m
for o in '{}' '{"leaf_cols":48,"leaf_rows":96}' '{"leaf_cols":32,"leaf_rows":96}' '{"leaf_cols":48,"leaf_rows":128}' '{"leaf_cols":64,"leaf_rows":128}' '{"nd_leaf":128}' '{"nd_leaf":160,"leaf_cols":48,"leaf_rows":96}' '{"nd_leaf":64}' '{"relax_z2":0.2,"relax_z3":0.1}' '{"leaf_cols":24,"leaf_rows":64}'; do
  timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --chol-opts "$o" 2>/dev/null
done | tee gpurun_out/r2m/sweep.log
