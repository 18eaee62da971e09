#!/bin/bash
# analysis-option sweep on the GPU box
for o in '{}' '{"leaf_cols":48,"leaf_rows":96}' '{"leaf_cols":24,"leaf_rows":48}' '{"leaf_cols":32,"leaf_rows":64,"nd_leaf":200}' '{"leaf_cols":32,"leaf_rows":64,"nd_leaf":48}' '{"leaf_cols":32,"leaf_rows":64,"relax_z2":0.3,"relax_z3":0.15}' '{"leaf_cols":32,"leaf_rows":64,"relax_small":16,"relax_z1":0.9,"relax_z2":0.5,"relax_z3":0.3}' '{"leaf_cols":16,"leaf_rows":64}'; do
  timeout -k 5 120 python bench.py --quick --steps 10 --warmup 2 --chol-opts "$o" 2>/dev/null
done
