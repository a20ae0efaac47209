#!/usr/bin/env python3
"""Condense gpurun_out/prof_sim_<tag>/ (scripts/collect_similarity_profile.sh) into profiles/<tag>_similarity_*.{csv,json}:
the kernels of one dlc_sdav_similarity_matrix call at 1063 x 30 x 2500 and the counters of its int8 product kernel."""
import collections, csv, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02i"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_sim_" + tag)
dst = os.path.join(root, "profiles")
KERNELS = ("gram_i8_kernel", "pair_score_amin_kernel", "sim_rows_kernel", "sim_range_kernel", "sim_colrange_kernel",
           "sim_keys_init_kernel", "sim_sample_kernel", "sim_pairwise_program_kernel", "distinctive_score_dma_kernel", "distinctive_score_kernel",
           "fill_diag_kernel", "gram_blocks_kernel", "sim_finish_kernel")
with open(os.path.join(src, "stats", "sim_kernel_stats.csv")) as f, open(os.path.join(dst, tag + "_similarity_kernel_stats.csv"), "w") as g:
    for i, line in enumerate(f):
        if i == 0 or any(k in line for k in KERNELS):
            g.write(line)
import subprocess
out = {"tag": tag, "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip(),
       "command": "rocprofv3 --kernel-trace [--stats | --pmc ... (separate passes)] -- python3 scripts/prof_similarity.py",
       "kernels": {}}
for row in csv.DictReader(open(os.path.join(src, "stats", "sim_kernel_stats.csv"))):
    for k in KERNELS:
        if k + "(" in row["Name"] or k + "<" in row["Name"]:
            out["kernels"][k] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    path = os.path.join(src, sub, "sim_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        for k in ("gram_i8_kernel", "pair_score_amin_kernel"):
            if k + "(" in r["Kernel_Name"]:
                pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in pmc:
    per = {c: v / len(disp[(k, c)]) for c, v in pmc[k].items()}
    e = out["kernels"].setdefault(k, {})
    e["pmc_per_dispatch"] = per
    if "GRBM_GUI_ACTIVE" in per and "avg_ns" in e:
        e["effective_clock_ghz"] = per["GRBM_GUI_ACTIVE"] / 8 / e["avg_ns"]          # summed over the 8 XCDs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "GRBM_GUI_ACTIVE" in per:
        e["mfma_busy_frac"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (per["GRBM_GUI_ACTIVE"] / 8)     # cycles per SIMD / kernel cycles
    if "SQ_WAVE_CYCLES" in per:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in per:
                e[c.lower() + "_frac_of_wave_cycles"] = per[c] / per["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in per:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reports 1/2 of a wide coalesced stream on gfx950 -> x 2 (fabric
        # requests of the L2, Infinity Cache hits included)
        e["fabric_read_bytes_corrected"] = per["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in per:
        e["fabric_write_bytes"] = per["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
        e["l2_hit_frac"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
json.dump(out, open(os.path.join(dst, tag + "_similarity_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
