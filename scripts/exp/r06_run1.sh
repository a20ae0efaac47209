mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_widths or create_rejects or unit or norm" > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu -k "json_contract or first_contact" > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_sim_r06; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for kind in saturated real; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$kind -o sim -- python3 $R/scripts/prof_similarity.py $kind > $OUT/stats_$kind.log 2>&1
grep "ms per call" $OUT/stats_$kind.log
done
