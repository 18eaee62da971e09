#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel for one bench workload (separate passes): gpurun_out/r03/pmc_<tag>.txt
W=${1:-stencil21}; TAG=${2:-$W}; STEPS=${3:-2}
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/r03/pm_$C
  timeout -k 10 500 rocprofv3 --pmc $C -d gpurun_out/r03/pm_$C -o p --output-format csv -- python3 bench.py --quick --steps $STEPS --warmup 1 --workload $W $BENCH_ARGS > gpurun_out/r03/pmc_$C.log 2>&1
done
python3 - <<PY > gpurun_out/r03/pmc_$TAG.txt
import csv, glob, re
tot = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/r03/pm_%s/**/*counter_collection.csv" % C, recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != C: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
        d = tot.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
        d[C] += float(r["Counter_Value"]); d["n"] += (C == "FETCH_SIZE")
steps = $STEPS + 1
print("# per step (%d steps in the run), raw counter KB -> MB; FETCH_SIZE x2 would apply to 16-B-per-lane streams only" % steps)
for k, d in sorted(tot.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    print("%-40s calls/step %7.1f  fetch %10.1f MB  write %10.1f MB" % (k[:40], d["n"] / steps, d["FETCH_SIZE"] / steps / 1024, d["WRITE_SIZE"] / steps / 1024))
PY
rm -rf gpurun_out/r03/pm_FETCH_SIZE gpurun_out/r03/pm_WRITE_SIZE
cat gpurun_out/r03/pmc_$TAG.txt | head -30
