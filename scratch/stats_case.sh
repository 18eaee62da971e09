#!/bin/bash
# rocprofv3 kernel stats of one bench_extra case: stats_case.sh <case> <tag>
CASE=${1:-lap3d}; TAG=${2:-lap3d}
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s2/stats_$TAG -o s --output-format csv -- python3 bench_extra.py --cases $CASE > gpurun_out/s2/stats_$TAG.log 2>&1
f=$(find gpurun_out/s2/stats_$TAG -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    nm = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("kvx::", "")
    print("%-28s calls %5s total %9.3f ms avg %9.1f us  %5.1f %%" % (nm[:28], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -3 gpurun_out/s2/stats_$TAG.log | cut -c1-600
rm -rf gpurun_out/s2/stats_$TAG
