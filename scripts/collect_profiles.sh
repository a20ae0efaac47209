#!/bin/bash
# GPU box only: rocprofv3 kernel-trace stats + PMC passes of the default bench (N=1).
# Output under gpurun_out/prof_<tag>/ ; summarise with scripts/summarize_profiles.py.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel stats: the whole default workload (headline loop + the `paths` rows); PMC passes: the headline loop only
S="python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-power-probe --no-shard-emulation --no-configs --no-rccl-smoke"
B="python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-power-probe --no-shard-emulation --no-configs --no-paths --no-rccl-smoke"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $S > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
ls $OUT
