#!/bin/bash
# kernel timeline of the last factor+solve of a bench_extra case: trace_case.sh <case> <tag> [ENV=val ...]
CASE=$1; TAG=$2; shift 2
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/s2/tr_$TAG -o t --output-format csv -- python3 bench_extra.py --cases $CASE --steps 6 > gpurun_out/s2/tr_$TAG.log 2>&1
f=$(find gpurun_out/s2/tr_$TAG -name "*kernel_trace.csv" | head -1)
python3 scratch/trace_timeline.py $f > gpurun_out/s2/timeline_$TAG.txt
rm -rf gpurun_out/s2/tr_$TAG
wc -l gpurun_out/s2/timeline_$TAG.txt
