#!/usr/bin/env python3
"""Condense gpurun_out/prof_fe_<tag>/ (scripts/collect_frontend_profile.sh) into profiles/<tag>_frontend_*.{csv,json}: the
kernels of the patch front-end (1063 frames of 192 x 240) and of the streaming cosine detector (1063 frames, batches of 32),
with the counters of the front-end's kernels per dispatch."""
import collections, csv, json, os, re, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_fe_" + tag)
dst = os.path.join(root, "profiles")
KERNELS = ("rgb_to_gray_kernel", "harris_candidates_kernel", "harris_select_kernel", "extract_patches_kernel",
           "small_topk_kernel", "score_gemm_kernel", "l2_normalize_regs_kernel", "exhaustive_topk_kernel")
with open(os.path.join(src, "stats", "fe_kernel_stats.csv")) as f, open(os.path.join(dst, tag + "_frontend_kernel_stats.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or any(k in line for k in KERNELS):
            g.write(line)
out = {"tag": tag, "command": "rocprofv3 --kernel-trace [--stats | --pmc ... (separate passes)] -- python3 scripts/prof_frontend.py",
       "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip(),
       "workload": "CvInputParser.parse_batch of 1063 uint8 frames 192 x 240 x 3 (noise, tiled real frames): 30 patches of "
                   "41 x 41 each; LoopClosureDetector.query_and_insert over 1063 frames of 4096-d in batches of 32",
       "log": [l.strip() for l in open(os.path.join(src, "stats.log")) if " ms" in l and "rocprof" not in l],
       "kernels": {}}
for row in csv.DictReader(open(os.path.join(src, "stats", "fe_kernel_stats.csv"))):
    m = re.search("|".join(KERNELS), row["Name"])
    if m:
        out["kernels"][m.group(0)] = {"calls": int(row["Calls"]), "avg_us": float(row["AverageNs"]) / 1e3,
                                     "min_us": float(row["MinNs"]) / 1e3, "max_us": float(row["MaxNs"]) / 1e3}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    path = os.path.join(src, sub, "fe_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        m = re.search("|".join(KERNELS[:4]), r["Kernel_Name"])
        if m:
            acc[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
            n[m.group(0)].add(r["Dispatch_Id"])
    for k, c in acc.items():
        d = out["kernels"].setdefault(k, {}).setdefault("pmc_per_dispatch", {})
        for name, v in c.items():
            d[name] = v / max(1, len(n[k]))
k = out["kernels"].get("harris_candidates_kernel", {})
p = k.get("pmc_per_dispatch", {})
if "SQ_INSTS_VALU" in p and "avg_us" in k:
    # a wave64 VALU instruction holds its SIMD16 for 4 cycles; 256 CUs x 4 SIMDs
    k["valu_issue_us_at_2p4GHz"] = p["SQ_INSTS_VALU"] * 4.0 / (256 * 4) / 2.4e3
    k["note"] = "VALU issue time of the kernel's instruction count against its measured duration: the kernel is VALU-bound"
for name, kk in out["kernels"].items():
    p = kk.get("pmc_per_dispatch", {})
    if "FETCH_SIZE" in p:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reports 1/2 of a wide coalesced stream on gfx950 -> x2.  The Harris
        # kernel's reads are 44-byte row segments of a tile (1.56 x the image by the halo, one or two 64-B lines each): its
        # raw figure already is ~2.8 x the 49 MB image, so both are kept
        kk["fetch_size_raw_bytes_per_dispatch"] = p["FETCH_SIZE"] * 1024
        kk["hbm_read_bytes_per_dispatch"] = p["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in p:
        kk["hbm_write_bytes_per_dispatch"] = p["WRITE_SIZE"] * 1024
json.dump(out, open(os.path.join(dst, tag + "_frontend_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
