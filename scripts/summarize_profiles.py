#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/collect_profiles.sh) into profiles/<tag>_*.{csv,json}."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
OURS = ("score_gemm_kernel", "finish_topk_kernel", "merge_topk_kernel", "l2_normalize_kernel")

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
with open(stats) as f, open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or any(k in line for k in OURS):
            g.write(line)

summary = {"tag": tag, "command": "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
           "kernels": {}}
for row in csv.DictReader(open(stats)):
    for k in OURS:
        if k in row["Name"]:
            summary["kernels"].setdefault(k, {})["avg_ns"] = float(row["AverageNs"])
            summary["kernels"][k]["calls"] = int(row["Calls"])
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    if not glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        continue
    f = newest(os.path.join(src, sub, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in OURS:
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        v = v[len(v) // 3:] or v                      # skip warm-up launches
        summary["kernels"].setdefault(k, {})[c] = sum(v) / len(v)
g = summary["kernels"].get("score_gemm_kernel", {})
if "FETCH_SIZE" in g:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reports 1/2 of a wide coalesced stream on gfx950 -> x2;
    # WRITE_SIZE (KiB) is exact for 16-B-per-lane stores (ours are 8-B float2 / 4-B: uncalibrated, small)
    g["hbm_read_bytes_per_launch_corrected"] = g["FETCH_SIZE"] * 1024 * 2
    g["hbm_write_bytes_per_launch"] = g.get("WRITE_SIZE", 0.0) * 1024
    g["hbm_traffic_bytes_per_launch"] = g["hbm_read_bytes_per_launch_corrected"] + g["hbm_write_bytes_per_launch"]
    g["workload"] = {"db_rows": 1000000, "dim": 4096, "queries": 256, "dtype": "bf16", "n_gpus": 1}
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary["kernels"].get("score_gemm_kernel", {}), indent=1))
