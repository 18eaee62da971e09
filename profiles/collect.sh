#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01b
# 1. bench.py JSON line, 2. rocprofv3 --kernel-trace --stats of the same command, 3./4. two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE: the TCC block cannot hold both, MI355X_MICROARCH.md).  Summaries land in gpurun_out/<tag>_*;
# profiles/summarize.py turns them into the small files committed under profiles/.
set -e -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ipm > $OUT/${TAG}_stats.log 2>&1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ipm > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ipm > $OUT/${TAG}_pmc_write.log 2>&1
python3 profiles/summarize.py $TAG
ls $OUT | grep $TAG
