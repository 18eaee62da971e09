#!/bin/bash
# wave-cycle breakdown of the kernels of a bench step: parked (s_waitcnt / barrier), issue-stalled, issuing
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -d gpurun_out/s2/pmc_sq -o p --output-format csv -- python3 bench.py --quick --steps 2 --warmup 1 > gpurun_out/s2/pmc_sq.log 2>&1
f=$(find gpurun_out/s2/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    agg[nm][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-26s %10s %7s %7s %7s %7s %7s %9s" % ("kernel", "wave_cyc", "parked", "istall", "active", "valu", "lds", "valu/wave"))
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]:
    wc = max(a["SQ_WAVE_CYCLES"], 1)
    print("%-26s %10.3g %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%% %9.0f" % (nm[:26], wc, 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc,
          100 * a["SQ_ACTIVE_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_VALU"] / wc, 100 * a["SQ_ACTIVE_INST_LDS"] / wc, a["SQ_INSTS_VALU"] / max(a["SQ_WAVES"], 1)))
PY
tail -2 gpurun_out/s2/pmc_sq.log | cut -c1-300
rm -rf gpurun_out/s2/pmc_sq
