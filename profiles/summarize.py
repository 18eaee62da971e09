#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of profiles/collect.sh into the small, committed files under profiles/:
   <tag>_bench.json               the bench.py line of the same build
   <tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average)
   <tag>_pmc_fetch_write_per_kernel.json   FETCH_SIZE / WRITE_SIZE (KB, raw counter values) per kernel and per step,
                                  plus per bench.py kernel family (bench.py reads "family_bytes_per_step")."""
import csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "").replace("kvx::", "")
    return re.sub(r"<.*", "", n)


FAMILY = {"k_scatter_a": "scatter_a", "k_front_wave": "front_small", "k_front_lds": "front_small", "k_assemble_big": "assemble_big",
          "k_potrf_blk": "potrf_diag", "k_trsm_blk": "trsm_panel", "k_syrk_trailing": "syrk_trailing",
          "k_fwd_wave": "fwd_level", "k_fwd_subtree": "fwd_level", "k_bwd_subtree": "bwd_level", "k_fwd_lds": "fwd_level", "k_fwd_big_init": "fwd_level", "k_fwd_big_step": "fwd_level",
          "k_bwd_wave": "bwd_level", "k_bwd_lds": "bwd_level", "k_bwd_big_init": "bwd_level", "k_bwd_big_step": "bwd_level"}

# bench line
b = os.path.join(out, tag + "_bench.json")
if os.path.exists(b):
    line = [l for l in open(b) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(prof, tag + "_bench.json"), "w"), indent=1)

# kernel stats
st = glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline  (12 factor+solve steps + 27 in the per-family timing loop)\n")
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("kvx::", "").replace(",", ";"),
                                              r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))

# PMC passes
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

fa, fc = pmc(tag + "_pmc_fetch", "FETCH_SIZE")
wa, wc = pmc(tag + "_pmc_write", "WRITE_SIZE")
if fa or wa:
    steps = 1 + 2 + 3 * 9          # warmup + timed + the per-family event-timing loop of bench.py (3 steps x 9 families)
    per = {}
    for k in sorted(set(fa) | set(wa)):
        per[k] = {"dispatches": round(fc.get(k, wc.get(k, 0)) / steps, 1), "FETCH_SIZE_KB": round(fa.get(k, 0.0) / steps, 1),
                  "WRITE_SIZE_KB": round(wa.get(k, 0.0) / steps, 1)}
    fam = {}
    for k, v in per.items():
        if k in FAMILY:
            fam[FAMILY[k]] = fam.get(FAMILY[k], 0.0) + (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0
    json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 "
                        "--no-cpu-baseline` (%d factor+solve steps in the process). KB per step, raw counter values: "
                        "MI355X_MICROARCH.md says FETCH_SIZE under-reports 16-B-per-lane streaming reads by 2x on gfx950 and other "
                        "widths are uncalibrated; these kernels issue 8-B-per-lane loads, so no correction is applied and the numbers "
                        "are read as a check against re-reads (traffic >> algorithmic bytes), not as absolutes. The syrk_trailing "
                        "family includes the fused next-diagonal-block factorisation." % steps,
               "steps_in_run": steps, "per_step": per, "family_bytes_per_step": fam},
              open(os.path.join(prof, tag + "_pmc_fetch_write_per_kernel.json"), "w"), indent=1)
print("summarised", tag)
