#!/bin/bash
# GPU box only: time the score GEMM of the shipped library and of the cache-policy builds (exp_build/lib_a_*.so),
# then one rocprofv3 PMC pass (FETCH_SIZE) per library for the fabric read traffic per launch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/exp_policy
mkdir -p $OUT
LIBS=$(ls $R/exp_build/lib_a_*.so | tr '\n' ':')
DLC_EXP_LIBS=$LIBS DLC_EXP_ROUNDS=3 python3 $R/scripts/exp_gemm.py > $OUT/timing.txt 2>&1
tail -20 $OUT/timing.txt
cd /tmp && export TMPDIR=/tmp
for lib in $R/deeploopcloser_amd/libdlc_hip.so $R/exp_build/lib_a_*.so; do
  name=$(basename $lib .so)
  DLC_EXP_LIBS=$lib DLC_EXP_SKIP_SHIPPED=1 DLC_EXP_ROUNDS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$name -- python3 $R/scripts/exp_gemm.py > $OUT/pmc_$name.log 2>&1
  f=$(ls $OUT/pmc_$name/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "score_gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = v[3:] or v
print("%-20s FETCH_SIZE x2 = %.3f GB per launch (%d launches)" % (sys.argv[2], sum(v) / len(v) * 1024 * 2 / 1e9, len(v)))
PY
done 2>&1 | tee $OUT/traffic.txt
