#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of profiles/collect_ipm.sh into profiles/<tag>_ipm_kernel_stats.csv (per kernel: calls,
total, average) and profiles/<tag>_ipm_pmc.json (FETCH_SIZE / WRITE_SIZE per launch of the KKT-layer kernels in bytes -- counter
values are KB, and FETCH_SIZE is scaled by the factor profiles/calibrate_fetch.py measured for 8-byte-per-lane reads (2.0, see
profiles/<tag>_pmc_fetch_write_per_kernel.json) -- next to their algorithmic bytes; bench.py reads "k_atda_bytes_per_launch":
the two launches of the assembly, k_atda_scale + k_atda)."""
import csv, glob, json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "").replace("kvx::", "")
    return re.sub(r"<.*", "", n)


st = glob.glob(os.path.join(out, tag + "_ipm_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(prof, tag + "_ipm_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/ipm_loop.py 3  (conelp, grid LP 250 x 200: ml = 200 000, n = 50 000; three full runs, the first on a new structure)\n")
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("kvx::", "").replace(",", ";"),
                                              r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))


def pmc(dirname, counter):
    f = glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True)
    acc, cnt = {}, {}
    if not f:
        return acc, cnt
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
        cnt[k] = cnt.get(k, 0) + 1
    return acc, cnt


fetch_scale = 2.0
try:
    fetch_scale = float(json.load(open(os.path.join(prof, tag + "_pmc_fetch_write_per_kernel.json")))["calibration"]["fetch_scale"])
except Exception:
    pass
fa, fc = pmc(tag + "_ipm_fetch", "FETCH_SIZE")
wa, wc = pmc(tag + "_ipm_write", "WRITE_SIZE")
if fa or wa:
    # FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3 (MI355X_MICROARCH.md, HBM section)
    ml, n, nnzG = 200000, 50000, 400000
    per = {}
    for k in sorted(set(fa) | set(wa)):
        if not (k.startswith("k_") and not k.startswith(("k_front", "k_fwd", "k_bwd", "k_potrf", "k_trsm", "k_syrk", "k_assemble", "k_scatter", "k_perm"))):
            continue
        launches = max(fc.get(k, 0), wc.get(k, 0), 1)
        per[k] = {"launches": launches, "fetch_bytes_per_launch": fa.get(k, 0.0) * 1024.0 * fetch_scale / launches,
                  "write_bytes_per_launch": wa.get(k, 0.0) * 1024.0 / launches}
        per[k]["traffic_bytes_per_launch"] = per[k]["fetch_bytes_per_launch"] + per[k]["write_bytes_per_launch"]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/ipm_loop.py 1",
           "units": "bytes (counter values are KB: x 1024; FETCH_SIZE x fetch_scale, the calibration of 8-byte-per-lane reads)",
           "fetch_scale": fetch_scale, "kernels": per}
    if "k_atda" in per:
        res["k_atda_bytes_per_launch"] = per["k_atda"]["traffic_bytes_per_launch"] + per.get("k_atda_scale", {}).get("traffic_bytes_per_launch", 0.0)
        res["k_atda_algorithmic_bytes_per_launch"] = 12.0 * nnzG + 8.0 * ml + 8.0 * 149550
    json.dump(res, open(os.path.join(prof, tag + "_ipm_pmc.json"), "w"), indent=1)
print("summarised", tag)
