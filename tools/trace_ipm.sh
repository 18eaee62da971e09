#!/bin/bash
# kernel timeline of one interior-point iteration of the bench's IPM leg: gpurun_out/r03/ipm_timeline${TAG}.txt
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/ipm_loop.py 3 > gpurun_out/r03/ipm_plain${TAG}.log 2>&1
rm -rf gpurun_out/r03/tri
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/r03/tri -o t --output-format csv -- python3 tools/ipm_loop.py 2 > gpurun_out/r03/ipm_trace${TAG}.log 2>&1
f=$(find gpurun_out/r03/tri -name "*kernel_trace.csv" | head -1)
python3 tools/ipm_timeline.py $f > gpurun_out/r03/ipm_timeline${TAG}.txt
rm -rf gpurun_out/r03/tri
cat gpurun_out/r03/ipm_plain${TAG}.log; tail -2 gpurun_out/r03/ipm_trace${TAG}.log; head -1 gpurun_out/r03/ipm_timeline${TAG}.txt
