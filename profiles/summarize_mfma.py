"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE):
    python3 profiles/summarize_mfma.py gpurun_out/<tag>_pmc_mfma/p_counter_collection.csv profiles/<tag>_mfma_util.json
SQ_VALU_MFMA_BUSY_CYCLES sums cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, PMC notes):
    mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024);  at 100 % the FP64 16x16x4 MFMA pipe delivers the 78.6 TF/s peak."""
import collections, csv, json, re, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur, seen = collections.Counter(), collections.Counter(), set()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    nm = m.group(1) if m else r["Kernel_Name"][:25]
    agg[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[nm] += 1; dur[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench_extra.py --cases lap3d",
       "workload": "7-point Laplacian 100^3 (n = 1e6), 4 factorisations + solves, kernels serialised by the counter collection",
       "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); FP64 MFMA peak 78.6 TF/s at 1.0",
       "kernels": {}}
for nm, _ in sorted(dur.items(), key=lambda kv: -kv[1])[:8]:
    a = agg[nm]
    util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["GRBM_GUI_ACTIVE"] / 8 * 1024, 1)
    out["kernels"][nm] = {"calls": cnt[nm], "ms_total": dur[nm] / 1e6, "mfma_busy_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"],
                          "grbm_gui_active": a["GRBM_GUI_ACTIVE"], "mfma_util": round(util, 4), "implied_tflops": round(util * 78.6, 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: (v["mfma_util"], v["implied_tflops"]) for k, v in out["kernels"].items()}))
