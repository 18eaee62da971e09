#!/bin/bash
# rocprofv3 --kernel-trace --stats of "python3 $@": prints the top kernels; keeps gpurun_out/r03/stats_<TAG>.csv
TAG=${TAG:-x}
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03/st_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/st_$TAG -o s --output-format csv -- python3 "$@" > gpurun_out/r03/st_$TAG.log 2>&1
f=$(find gpurun_out/r03/st_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/r03/stats_$TAG.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,pct\n")
    for r in rows:
        nm = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("kvx::", "").replace(",", ";")
        o.write("%s,%s,%.1f,%.2f,%s\n" % (nm, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
for l in open(sys.argv[2]).read().splitlines()[:24]:
    print(l)
PY
rm -rf gpurun_out/r03/st_$TAG
