#!/bin/bash
# Collect the judged profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r04
# 1. bench.py JSON line (the whole default run), 2. rocprofv3 --kernel-trace --stats of the same command without its CPU legs,
# 3./4. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: the TCC block cannot hold both, MI355X_MICROARCH.md) of the headline
# system and of the 21-point system, 5. the same two counters on a kernel of known byte count in the library's own access pattern
# (profiles/calibrate_fetch.py).  Summaries land in gpurun_out/<tag>_*; profiles/summarize.py turns them into the small files
# committed under profiles/.  The program goes directly after `--` (no env / bash -c hop under the profiler).
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
LEAN="--no-cpu-baseline --no-ipm --no-klu --no-extra --no-one-shot"
timeout -k 10 900 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 400 $OUT/${TAG}_bench.json; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 $LEAN > $OUT/${TAG}_stats.log 2>&1
echo "stats done"
for W in lap2d stencil21; do
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_fetch_$W -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --quick --workload $W > $OUT/${TAG}_pmc_fetch_$W.log 2>&1
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_write_$W -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --quick --workload $W > $OUT/${TAG}_pmc_write_$W.log 2>&1
  echo "pmc $W done"
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_cal_fetch -o p --output-format csv -- python3 profiles/calibrate_fetch.py > $OUT/${TAG}_cal_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_cal_write -o p --output-format csv -- python3 profiles/calibrate_fetch.py > $OUT/${TAG}_cal_write.log 2>&1
# the 21-point system's kernel stats (the system the trailing update is judged on) and the FP64 ceilings of the chip
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats21 -o s --output-format csv -- python3 bench.py --steps 5 --warmup 2 --quick --workload stencil21 > $OUT/${TAG}_stats21.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fp64_peak.hip -o /tmp/fp64_peak 2>/dev/null && timeout -k 5 120 /tmp/fp64_peak > $OUT/${TAG}_fp64_peak.txt 2>&1
python3 profiles/summarize.py $TAG
ls $OUT | grep $TAG | head -30
