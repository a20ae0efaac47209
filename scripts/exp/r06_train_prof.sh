R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_train_r06; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python3 $R/scripts/prof_train_replay.py > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, os, collections
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_train_r06/tr_kernel_trace.csv")
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 steps: find steps by the random_mask kernel
idx = [i for i, r in enumerate(rows) if "random_mask" in r["Kernel_Name"]]
lo, hi = idx[-20], idx[-1]
agg = collections.OrderedDict()
steps = len([i for i in idx if lo <= i < hi])
for r in rows[lo:hi]:
    n = r["Kernel_Name"].split("(")[0][-60:] + " g" + r["Grid_Size_X"]
    agg.setdefault(n, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in agg.items():
    print("%-75s x%.1f/step  %.1f us each  %.1f us/step" % (k, len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / steps / 1e3)); tot += sum(v) / steps / 1e3
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / steps / 1e3
print("kernel sum %.1f us/step, step span %.1f us" % (tot, span))
PY
