#!/bin/bash
# rocprofv3 kernel stats of CnnVtl.transform (GPU box only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_cnn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/prof_cnn.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
f=$(ls -t $OUT/*/*kernel_stats.csv | head -1)
cut -c1-160 $f | head -14
