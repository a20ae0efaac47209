#!/usr/bin/env python3
"""Condense gpurun_out/prof_split_<tag>/ (scripts/collect_split_profile.sh) into profiles/<tag>_split_*.{csv,json}: the kernels
of SDAV.transform in the tolerance mode (1063 frames) and the counters of its fp16 MFMA GEMM."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_split_" + tag)
dst = os.path.join(root, "profiles")
KERNELS = ("gemm_split_f16_kernel", "sp_split_rows_kernel", "sp_split_weights_kernel", "sp_absmax_kernel", "sp_scale_kernel")
with open(os.path.join(src, "stats", "sp_kernel_stats.csv")) as f, open(os.path.join(dst, tag + "_split_kernel_stats.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or any(k in line for k in KERNELS):
            g.write(line)
import subprocess
out = {"tag": tag, "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip(),
       "command": "rocprofv3 --kernel-trace [--stats | --pmc ... (separate passes)] -- python3 scripts/prof_sdav_split.py",
       "workload": "SDAV.transform(dtype='f16x2') of 1063 frames: 4 launches of gemm_split_f16_kernel<false> + 1 of <true> per call",
       "kernels": {}}
for row in csv.DictReader(open(os.path.join(src, "stats", "sp_kernel_stats.csv"))):
    for k in KERNELS:
        if k in row["Name"]:
            name = k + ("<final>" if "<true>" in row["Name"] else ("<hidden>" if "<false>" in row["Name"] else ""))
            out["kernels"][name] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    path = os.path.join(src, sub, "sp_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if "gemm_split_f16_kernel<false>" in r["Kernel_Name"]:
            pmc[sub][r["Counter_Name"]] += float(r["Counter_Value"])
            n[sub].add(r["Dispatch_Id"])
k = out["kernels"].get("gemm_split_f16_kernel<hidden>")
if k:
    c = {}
    for sub in pmc:
        for name, v in pmc[sub].items():
            c[name] = v / max(1, len(n[sub]))
    k["pmc_per_dispatch"] = c
    t = k["avg_ns"] * 1e-9
    if "GRBM_GUI_ACTIVE" in c:
        k["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        k["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * c["GRBM_GUI_ACTIVE"] / 8)
    if "FETCH_SIZE" in c:
        k["fabric_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        k["fabric_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in c:
        k["l2_hit_frac"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if w in c and "SQ_WAVE_CYCLES" in c:
            k[w.lower() + "_frac_of_wave_cycles"] = c[w] / c["SQ_WAVE_CYCLES"]
json.dump(out, open(os.path.join(dst, tag + "_split_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out["kernels"], indent=1))
