#!/bin/bash
# kernel trace of a few bench steps -> gpurun_out/s2/trace_<tag>/
TAG=${1:-v2}
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/s2/trace_$TAG -o t --output-format csv -- python3 bench.py --quick --steps 6 --warmup 3 > gpurun_out/s2/trace_$TAG.log 2>&1
f=$(find gpurun_out/s2/trace_$TAG -name "*kernel_trace.csv" | head -1)
python3 scratch/trace_timeline.py $f > gpurun_out/s2/timeline_$TAG.txt
rm -rf gpurun_out/s2/trace_$TAG
wc -l gpurun_out/s2/timeline_$TAG.txt
