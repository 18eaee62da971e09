#!/bin/bash
mkdir -p gpurun_out/r2m
for z3 in 0.05 0.075 0.1 0.125; do for z2 in 0.1 0.125 0.15 0.175 0.2; do
  o="{\"relax_z2\":$z2,\"relax_z3\":$z3}"
  timeout -k 5 120 python bench.py --quick --steps 20 --warmup 4 --chol-opts "$o" 2>/dev/null
done; done | tee gpurun_out/r2m/sweep4.log
