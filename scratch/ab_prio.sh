#!/bin/bash
set -e
mkdir -p gpurun_out/r2d
rm -f gpurun_out/r2d/ab2.log
for P in 0 1; do for G in 0 1; do
  echo "KVX_CHAIN_PRIO=$P KVX_NO_GRAPH=$G" >> gpurun_out/r2d/ab2.log
  KVX_MID_M=0 KVX_CHAIN_PRIO=$P KVX_NO_GRAPH=$G timeout -k 10 120 python bench.py --quick --steps 20 --warmup 3 >> gpurun_out/r2d/ab2.log 2>> gpurun_out/r2d/ab2.err
done; done
cat gpurun_out/r2d/ab2.log
