#!/bin/bash
# kernel timeline of the last factor+solve step of bench.py (graph replay on): gpurun_out/r03/timeline.txt
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03/tr
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/r03/tr -o t --output-format csv -- python3 bench.py --quick --steps 6 --warmup 4 $BENCH_ARGS > gpurun_out/r03/trace.log 2>&1
f=$(find gpurun_out/r03/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f > gpurun_out/r03/timeline${TAG}.txt
rm -rf gpurun_out/r03/tr
tail -3 gpurun_out/r03/trace.log
