#!/bin/bash
# SQ counters of one bench_extra case: pmc_case.sh <case> "<counters>"
CASE=$1; CTRS=$2
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --pmc $CTRS -d gpurun_out/s2/pmc_case -o p --output-format csv -- python3 bench_extra.py --cases $CASE --steps 3 > gpurun_out/s2/pmc_case.log 2>&1
f=$(find gpurun_out/s2/pmc_case -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = []
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    agg[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
print("%-28s " % "kernel" + " ".join("%16s" % n[-16:] for n in names))
key = names[0]
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][key])[:6]:
    print("%-28s " % nm[:28] + " ".join("%16.4g" % a[n] for n in names))
PY
rm -rf gpurun_out/s2/pmc_case
