# r06: the bench's kernel stats + PMC passes (scripts/collect_profiles.sh), and the counters of configs[1]'s cosine top-20
R=$GRAFT_REPO_ROOT
bash $R/scripts/collect_profiles.sh r06 > $R/gpurun_out/collect_r06.log 2>&1; echo "collect rc=$?"
OUT=$R/gpurun_out/prof_cos_r06; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o cos -- python3 $R/scripts/prof_cos_topk.py > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE --output-format csv -d $OUT/pmc_tcc -o cos -- python3 $R/scripts/prof_cos_topk.py > $OUT/pmc_tcc.log 2>&1
ls $OUT $OUT/pmc_fetch | head -20
