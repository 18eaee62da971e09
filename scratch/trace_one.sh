#!/bin/bash
# kernel timeline of the LAST 64-rhs solve of scratch/multirhs.py one
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/s2/trace_one -o t --output-format csv -- python3 scratch/multirhs.py one > gpurun_out/s2/trace_one.log 2>&1
f=$(find gpurun_out/s2/trace_one -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, int(r.get("Stream_Id", 0) or 0)))
ks.sort()
# last solve: from the last k_perm_gather to the last k_perm_scatter
gi = max(i for i, k in enumerate(ks) if k[2].startswith("k_perm_gather"))
si = max(i for i, k in enumerate(ks) if k[2].startswith("k_perm_scatter"))
t0 = ks[gi][0]
print("solve span %.3f ms" % ((ks[si][1] - t0) / 1e6))
# group consecutive launches of the same kernel name
out = []
for s, e, nm, st in ks[gi:si + 1]:
    if out and out[-1][0] == nm and s - out[-1][2] < 200000:
        out[-1][2] = max(out[-1][2], e); out[-1][3] += 1; out[-1][4] += e - s
    else:
        out.append([nm, s, e, 1, e - s])
for nm, s, e, c, busy in out:
    print("%8.1f us  +%7.1f us  x%-3d busy %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, c, busy / 1e3, nm[:40]))
PY
rm -rf gpurun_out/s2/trace_one
