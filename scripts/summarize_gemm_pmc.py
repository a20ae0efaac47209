#!/usr/bin/env python3
"""Condense gpurun_out/prof_gemm_<tag>/ (scripts/collect_gemm_pmc.sh) into profiles/<tag>_gemm_f64_pmc.json and
profiles/<tag>_gemm_f64_kernel_stats.csv: per instantiation of gemm_bias_act_kernel the launch count, the mean
duration and the mean counter values per launch (plus the derived MFMA-busy share and L2 hit rate)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_gemm_" + tag)
dst = os.path.join(root, "profiles")
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
with open(stats) as f, open(os.path.join(dst, tag + "_gemm_f64_kernel_stats.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or "dlc" in line or "_kernel" in line:
            g.write(line)
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python3 scripts/prof_paths_gemm.py",
           "units": "SQ_* in quad-cycles summed over waves except SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (cycles); "
                    "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: double FETCH_SIZE for wide streams)",
           "kernels": {}}
for row in csv.DictReader(open(stats)):
    if "gemm_bias_act_kernel" in row["Name"] or "gemm_dma_f64_kernel" in row["Name"] or "pair_score" in row["Name"] or "distinctive" in row["Name"]:
        summary["kernels"].setdefault(row["Name"], {}).update(avg_ns=float(row["AverageNs"]), calls=int(row["Calls"]),
                                                               total_ns=float(row["TotalDurationNs"]))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    fs = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if r["Kernel_Name"] in summary["kernels"]:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        summary["kernels"][k][c] = sum(v) / len(v)
for k, v in summary["kernels"].items():
    if v.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # MFMA-busy cycles are summed over the chip's 1024 SIMDs; SQ_BUSY_CYCLES over its 32 shader engines' SQs
        v["mfma_busy_share_of_wave_time"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") else None
    if "TCC_HIT_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
json.dump(summary, open(os.path.join(dst, tag + "_gemm_f64_pmc.json"), "w"), indent=1, sort_keys=True)
for k, v in summary["kernels"].items():
    print(k[:90], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
