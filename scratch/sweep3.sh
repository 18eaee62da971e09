#!/bin/bash
mkdir -p gpurun_out/r2m
for o in '{}' '{"relax_z2":0.2,"relax_z3":0.1}' '{"relax_z2":0.3,"relax_z3":0.15}' '{"relax_z2":0.3,"relax_z3":0.1}' '{"relax_z2":0.2,"relax_z3":0.2}' '{"relax_z1":0.9,"relax_z2":0.4,"relax_z3":0.2}' '{"relax_z2":0.15,"relax_z3":0.075}' '{"relax_small":8,"relax_z2":0.2,"relax_z3":0.1}' '{"relax_small":16,"relax_z2":0.2,"relax_z3":0.1}' '{"relax_z2":0.25,"relax_z3":0.25}' '{"relax_z2":0.5,"relax_z3":0.3}'; do
  timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --chol-opts "$o" 2>/dev/null
done | tee gpurun_out/r2m/sweep3.log
