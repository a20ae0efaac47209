#!/bin/bash
# GPU box only: rocprofv3 kernel stats of scripts/bench_paths.py (encode / similarity / distance rows of SURVEY section 8).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_paths_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/bench_paths.py > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
grep '^{' $OUT/log.txt > $OUT/bench_paths.jsonl
ls $OUT
