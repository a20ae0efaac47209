"""Timeline of a rocprofv3 kernel trace (…_kernel_trace.csv): per kernel name the count and mean duration, and the idle time of
the device between consecutive kernels, grouped by the kernel that FOLLOWS the gap.   usage: trace_gaps.py trace.csv [last N kernels]"""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
prev_end = None
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|dlc_gemm::|void ", "", r["Kernel_Name"]).split("(")[0][:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n] += e - s; cnt[n] += 1
    if prev_end is not None:
        gap[n] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels %d, span %.1f us, busy %.1f us, idle %.1f us" % (len(rows), span / 1e3, sum(dur.values()) / 1e3, sum(gap.values()) / 1e3))
for n in sorted(dur, key=lambda k: -dur[k]):
    print("%-42s x%4d  mean %7.1f us   idle before it, mean %6.1f us" % (n, cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3))
