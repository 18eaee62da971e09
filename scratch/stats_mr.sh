#!/bin/bash
# rocprofv3 kernel stats of the 64-rhs solve at n = 1e6 (scratch/multirhs.py bigprof: 6 solves)
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s2/stats_mr -o s --output-format csv -- python3 scratch/multirhs.py bigprof > gpurun_out/s2/stats_mr.log 2>&1
f=$(find gpurun_out/s2/stats_mr -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys
tot = 0.0
for r in list(csv.DictReader(open(sys.argv[1])))[:30]:
    nm = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("kvx::", "")
    print("%-44s calls %5s total/6 %9.3f ms avg %9.1f us" % (nm[:44], r["Calls"], int(r["TotalDurationNs"]) / 6e6, float(r["AverageNs"]) / 1e3))
PY
tail -3 gpurun_out/s2/stats_mr.log | cut -c1-400
rm -rf gpurun_out/s2/stats_mr
