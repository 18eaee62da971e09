#!/bin/bash
# GPU busy fraction of the interior-point loop (config 4b): kernel trace of lp_trace.py, union of kernel intervals over the loop
mkdir -p gpurun_out/s2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/s2/tr_lp -o t --output-format csv -- python3 scratch/lp_trace.py > gpurun_out/s2/tr_lp.log 2>&1
f=$(find gpurun_out/s2/tr_lp -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys, re, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the second conelp call = last ~60 % of the trace: take the kernels after the last big gap (> 5 ms)
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if st[i] - en[i - 1] > 3_000_000: cut = i
rows, st, en = rows[cut:], st[cut:], en[cut:]
iv = sorted(zip(st, en))
busy, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for a, b in iv[1:]:
    if a > ce:
        busy += ce - cs; gaps.append(a - ce); cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
tot = iv[-1][1] - iv[0][0]
print("kernels %d, span %.2f ms, busy %.2f ms (%.1f %%)" % (len(rows), tot / 1e6, busy / 1e6, 100.0 * busy / tot))
big = sorted(gaps, reverse=True)[:60]
print("gaps > 20 us: %d, sum %.2f ms; top: %s" % (sum(g > 20000 for g in gaps), sum(g for g in gaps if g > 20000) / 1e6, [round(g / 1e3) for g in big[:25]]))
d = collections.Counter()
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("kvx::", "")
    d[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in d.most_common(12): print("  %-30s %.2f ms" % (k[:30], v / 1e6))
PY
tail -1 gpurun_out/s2/tr_lp.log
rm -rf gpurun_out/s2/tr_lp
