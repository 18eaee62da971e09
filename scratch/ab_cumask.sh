#!/bin/bash
set -e
mkdir -p gpurun_out/r2k
rm -f gpurun_out/r2k/ab.log
for M in 0 3 2; do for G in 0 1; do
  echo "KVX_SIDE_CU_MASK=$M KVX_NO_GRAPH=$G" >> gpurun_out/r2k/ab.log
  KVX_SIDE_CU_MASK=$M KVX_NO_GRAPH=$G timeout -k 10 120 python bench.py --quick --steps 20 --warmup 3 >> gpurun_out/r2k/ab.log 2>> gpurun_out/r2k/ab.err
done; done
cat gpurun_out/r2k/ab.log
