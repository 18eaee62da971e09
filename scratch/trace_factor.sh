#!/bin/bash
# kernel timeline of the LAST factorisation of bench.py --steps 3 (config 2)
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/s2/trace_f -o t --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-ipm > gpurun_out/s2/trace_f.log 2>&1
f=$(find gpurun_out/s2/trace_f -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm))
ks.sort()
ci = max(i for i, k in enumerate(ks) if k[2].startswith("k_clear_factor"))
gi = max(i for i, k in enumerate(ks) if k[2].startswith("k_perm_gather"))
t0 = ks[ci][0]
print("factor span %.3f ms" % ((ks[gi][0] - t0) / 1e6))
out = []
for s, e, nm in ks[ci:gi]:
    if out and out[-1][0] == nm and s - out[-1][2] < 100000:
        out[-1][2] = max(out[-1][2], e); out[-1][3] += 1; out[-1][4] += e - s
    else:
        out.append([nm, s, e, 1, e - s])
for nm, s, e, c, busy in out:
    print("%8.1f us  +%7.1f us  x%-3d busy %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, c, busy / 1e3, nm[:40]))
PY
rm -rf gpurun_out/s2/trace_f
