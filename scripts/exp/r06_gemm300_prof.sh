R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_gemm300; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o g -- python3 $R/scripts/exp/r06_gemm300.py > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, os, collections
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/prof_gemm300/g_kernel_trace.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" in n or "splitk" in n:
        agg[(n[:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print(k, len(v), "avg %.1f us" % (sum(v) / len(v) / 1e3))
PY
