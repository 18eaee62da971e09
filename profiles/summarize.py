#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of profiles/collect.sh into the small, committed files under profiles/:
   <tag>_bench.json               the bench.py line of the same build
   <tag>_kernel_stats.csv         rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average)
   <tag>_pmc_fetch_write_per_kernel.json   FETCH_SIZE / WRITE_SIZE per kernel and per factor+solve step for the headline system
                                  ("config2") and the 21-point system ("stencil21"), the calibration of the two counters on a kernel
                                  of known byte count in the library's access pattern (8 bytes per lane), the per-family bytes
                                  bench.py reads ("family_bytes_per_step", already scaled by the calibration) and the fingerprint of
                                  the library sources the profile was taken on."""
import csv, glob, hashlib, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def clean(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("kvx::", "")


def short(name):
    return re.sub(r"<.*", "", clean(name))


def source_fingerprint():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "kvxopt_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "kvxopt_amd", "csrc", "*.[ch]pp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


FAMILY = {"k_scatter_a": "scatter_a", "k_init_factor": "scatter_a", "k_front_wave": "front_small", "k_front_lds": "front_small", "k_assemble_big": "assemble_big",
          "k_potrf_blk": "potrf_diag", "k_trsm_blk": "trsm_panel", "k_syrk_trailing": "syrk_trailing", "k_syrk_trailing128": "syrk_trailing", "k_syrk_lds": "syrk_trailing",
          "k_assemble_big_potrf": "assemble_big",
          "k_fwd_wave": "fwd_level", "k_fwd_subtree": "fwd_level", "k_bwd_subtree": "bwd_level", "k_fwd_lds": "fwd_level", "k_fwd_big_step": "fwd_level",
          "k_bwd_wave": "bwd_level", "k_bwd_lds": "bwd_level", "k_bwd_big_init": "bwd_level", "k_bwd_big_step": "bwd_level"}

# bench line
b = os.path.join(out, tag + "_bench.json")
if os.path.exists(b):
    lines = [l for l in open(b) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(prof, tag + "_bench.json"), "w"), indent=1)

# kernel stats
st = glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    nfac = sum(int(r["Calls"]) for r in rows if clean(r["Name"]).startswith(("k_clear_factor", "k_init_factor")))
    with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-ipm --no-klu --no-extra --no-one-shot\n")
        f.write("# %d numeric factorisations in the process: 12 one-enqueue steps (factor + solve, forward sweep beside the top of the tree), 8 steps as two calls, "
                "24 in the per-family timing loop (graphs off)\n" % nfac)
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (clean(r["Name"]).replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))


st21 = glob.glob(os.path.join(out, tag + "_stats21", "**", "*kernel_stats.csv"), recursive=True)
if st21:
    rows = list(csv.DictReader(open(st21[0])))
    nfac = sum(int(r["Calls"]) for r in rows if clean(r["Name"]).startswith(("k_clear_factor", "k_init_factor")))
    with open(os.path.join(prof, tag + "_kernel_stats_stencil21.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --quick --workload stencil21  (%d numeric factorisations, one-enqueue steps)\n" % nfac)
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (clean(r["Name"]).replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
pk = os.path.join(out, tag + "_fp64_peak.txt")
if os.path.exists(pk):
    shutil.copy(pk, os.path.join(prof, tag + "_fp64_peak.txt"))
lu = glob.glob(os.path.join(out, tag + "_lustats", "**", "*kernel_stats.csv"), recursive=True)
if lu:
    rows = list(csv.DictReader(open(lu[0])))
    with open(os.path.join(prof, tag + "_lu_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/lu_prof.py  (kvxopt.klu on ACTIVSg2000: refactor + solve steps)\n")
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (clean(r["Name"]).replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
lu2 = glob.glob(os.path.join(out, tag + "_lu2dstats", "**", "*kernel_stats.csv"), recursive=True)
if lu2:
    rows = list(csv.DictReader(open(lu2[0])))
    with open(os.path.join(prof, tag + "_lu2d_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench_extra.py --cases lu2d  (600 x 600 convection-diffusion: first factorisation with its merge passes, then the timed refactor + solve steps)\n")
        f.write("kernel,calls,total_us,avg_us,pct\n")
        for r in rows:
            f.write("%s,%s,%.1f,%.2f,%s\n" % (clean(r["Name"]).replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
lj = os.path.join(out, tag + "_lu.jsonl")
if os.path.exists(lj):
    shutil.copy(lj, os.path.join(prof, tag + "_lu.jsonl"))


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


# calibration: k_scal, 2^26 doubles, 5 launches: 512 MiB read and written per launch, 8 bytes per lane
cf, cfc = pmc(tag + "_cal_fetch", "FETCH_SIZE")
cw, cwc = pmc(tag + "_cal_write", "WRITE_SIZE")
known = 8.0 * (1 << 26)
cal = {}
if cf.get("k_scal") and cw.get("k_scal"):
    cal = {"kernel": "k_scal (x := a x), 2^26 doubles, 8 bytes per lane", "known_bytes_read_per_launch": known, "known_bytes_written_per_launch": known,
           "FETCH_SIZE_per_launch_raw": cf["k_scal"] / cfc["k_scal"], "WRITE_SIZE_per_launch_raw": cw["k_scal"] / cwc["k_scal"]}
    # rocprofv3 reports both counters in KiB on this stack
    cal["fetch_scale"] = known / (cal["FETCH_SIZE_per_launch_raw"] * 1024.0)
    cal["write_scale"] = known / (cal["WRITE_SIZE_per_launch_raw"] * 1024.0)
fscale = cal.get("fetch_scale", 1.0)
wscale = cal.get("write_scale", 1.0)

res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 --quick [--workload W]`. "
                "Counter values are KiB; per_step = raw KiB per factor+solve step (steps counted from the k_init_factor launches of the run); "
                "family_bytes_per_step = (FETCH_SIZE x fetch_scale + WRITE_SIZE x write_scale) x 1024 per step, the scales from the calibration kernel "
                "(known byte count, the library's 8-byte-per-lane pattern) as MI355X_MICROARCH.md prescribes for access widths other than 16 bytes per lane.",
       "source_fingerprint": source_fingerprint(), "calibration": cal, "fetch_scale": round(fscale, 4), "write_scale": round(wscale, 4),
       "fetch_scale_note": "known bytes / counter on k_scal (8 B per lane streaming read)"}
for key, W in (("config2", "lap2d"), ("stencil21", "stencil21")):
    fa, fc = pmc("%s_pmc_fetch_%s" % (tag, W), "FETCH_SIZE")
    wa, wc = pmc("%s_pmc_write_%s" % (tag, W), "WRITE_SIZE")
    if not (fa or wa):
        continue
    steps = max(fc.get("k_init_factor", wc.get("k_init_factor", fc.get("k_clear_factor", wc.get("k_clear_factor", 1)))), 1)
    per = {}
    for k in sorted(set(fa) | set(wa)):
        per[k] = {"dispatches": round(fc.get(k, wc.get(k, 0)) / steps, 1), "FETCH_SIZE_KB": round(fa.get(k, 0.0) / steps, 1),
                  "WRITE_SIZE_KB": round(wa.get(k, 0.0) / steps, 1)}
    fam = {}
    for k, v in per.items():
        if k in FAMILY:
            fam[FAMILY[k]] = fam.get(FAMILY[k], 0.0) + (v["FETCH_SIZE_KB"] * fscale + v["WRITE_SIZE_KB"] * wscale) * 1024.0
    res[key] = {"steps_in_run": steps, "per_step": per, "family_bytes_per_step": fam}
if "config2" in res or "stencil21" in res:
    json.dump(res, open(os.path.join(prof, tag + "_pmc_fetch_write_per_kernel.json"), "w"), indent=1)
print("summarised", tag, "calibration:", {k: cal.get(k) for k in ("fetch_scale", "write_scale")})
