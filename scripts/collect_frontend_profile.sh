#!/bin/bash
# GPU box only: rocprofv3 kernel-trace stats + PMC passes of the patch front-end and the streaming cosine detector
# (scripts/prof_frontend.py).  Output under gpurun_out/prof_fe_<tag>/; summarise with scripts/summarize_frontend_profile.py.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
OUT=$R/gpurun_out/prof_fe_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S="python3 $R/scripts/prof_frontend.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o fe -- $S > $OUT/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq -o fe -- $S > $OUT/pmc_sq.log 2>&1
# (FETCH_SIZE and WRITE_SIZE in one pass: "Request exceeds the capabilities of the hardware to collect")
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o fe -- $S > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o fe -- $S > $OUT/pmc_write.log 2>&1
ls $OUT
