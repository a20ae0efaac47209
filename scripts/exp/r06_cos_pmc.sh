R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_cos_r06b; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -o cos -- python3 $R/scripts/prof_cos_topk.py > $OUT/pmc_sq.log 2>&1
tail -2 $OUT/pmc_sq.log
