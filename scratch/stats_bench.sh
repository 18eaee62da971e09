#!/bin/bash
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s2/stats_b -o s --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ipm > gpurun_out/s2/stats_b.log 2>&1
f=$(find gpurun_out/s2/stats_b -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    nm = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("kvx::", "")
    print("%-36s calls %5s total %9.3f ms avg %9.1f us  %5.1f %%" % (nm[:36], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 gpurun_out/s2/stats_b.log | cut -c1-300
rm -rf gpurun_out/s2/stats_b
