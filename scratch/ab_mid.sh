#!/bin/bash
# A/B of the one-workgroup-per-front kernel for mid-size fronts (KVX_MID_M) on config 2
set -e
mkdir -p gpurun_out/r2d
for M in 0 256 384 192; do
  echo "KVX_MID_M=$M" >> gpurun_out/r2d/ab.log
  KVX_MID_M=$M timeout -k 10 120 python bench.py --quick --steps 20 --warmup 3 >> gpurun_out/r2d/ab.log 2>> gpurun_out/r2d/ab.err
done
cat gpurun_out/r2d/ab.log
