#!/bin/bash
# kernel-level breakdown of a 64-rhs solve at n = 1e6 (run through gpurun from the repo root)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/mr_stats -o s --output-format csv -- python3 scratch/multirhs.py bigprof > $OUT/mr_stats.log 2>&1
f=$(ls $OUT/mr_stats/*kernel_stats.csv $OUT/mr_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
head -14 $f | cut -c1-160
