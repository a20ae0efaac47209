#!/bin/bash
# GPU box only: rocprofv3 kernel-trace stats + PMC passes of the SDAV similarity matrix at the reference's size
# (scripts/prof_similarity.py).  Output under gpurun_out/prof_sim_<tag>/; summarise with scripts/summarize_similarity_profile.py.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02i}
KIND=${2:-saturated}        # data kind of scripts/prof_similarity.py (saturated | real | real_fan_in | ...)
OUT=$R/gpurun_out/prof_sim_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S="python3 $R/scripts/prof_similarity.py $KIND"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o sim -- $S > $OUT/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o sim -- $S > $OUT/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o sim -- $S > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o sim -- $S > $OUT/pmc_write.log 2>&1
ls $OUT
