#!/bin/bash
# Profile artefacts of the KKT-layer kernels (kkt.hip: NT scaling, S = G' D G assembly, mat-vecs) inside the device-resident
# interior-point loop, BASELINE configs[3] in inequality form (config 4b).  Run through gpurun from the repo root:
#   bash profiles/collect_ipm.sh r03
# 1. rocprofv3 --kernel-trace --stats of three conelp runs; 2./3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: the
# TCC block cannot hold both).  The program goes directly after `--`.  profiles/summarize_ipm.py condenses them.
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ipm_stats -o s --output-format csv -- python3 tools/ipm_loop.py 3 > $OUT/${TAG}_ipm_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_ipm_fetch -o p --output-format csv -- python3 tools/ipm_loop.py 1 > $OUT/${TAG}_ipm_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_ipm_write -o p --output-format csv -- python3 tools/ipm_loop.py 1 > $OUT/${TAG}_ipm_write.log 2>&1
python3 profiles/summarize_ipm.py $TAG
ls $OUT | grep ${TAG}_ipm
