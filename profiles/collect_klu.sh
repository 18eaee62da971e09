#!/bin/bash
# Profile artefacts of the sparse-LU (kvxopt.klu) path, run through gpurun from the repo root:
#   bash profiles/collect_klu.sh r01c
# 1. bench_extra.py JSON lines (klu3 = BASELINE configs[2], lu2d = unsymmetric 600 x 600 grid), 2. rocprofv3 kernel
# stats of ten refactor+solve steps on ACTIVSg2000 (tools/lu_prof.py), 3. of the lu2d case of bench_extra.py.
set -e -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 10 600 python3 bench_extra.py --cases klu3,lu2d > $OUT/${TAG}_lu.jsonl 2> $OUT/${TAG}_lu.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_lustats -o s --output-format csv -- python3 tools/lu_prof.py > $OUT/${TAG}_lustats.log 2>&1
# 3. the same of the 600 x 600 convection-diffusion matrix (the blocked big-front chain: panel / interchanges / update)
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_lu2dstats -o s --output-format csv -- python3 bench_extra.py --cases lu2d > $OUT/${TAG}_lu2dstats.log 2>&1
ls $OUT | grep $TAG
