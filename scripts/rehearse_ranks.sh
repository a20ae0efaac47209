set -e
F="--steps 20 --warmup 5 --no-paths --no-cpu-baseline --no-power-probe --no-shard-emulation --no-configs"
for n in 1 2 4; do
  if [ $n = 1 ]; then X=""; else X="--backend gloo --share-gpu"; fi
  timeout -k 10 500 python bench.py --gpus $n $X $F > gpurun_out/reh_$n.json 2> gpurun_out/reh_$n.err
  timeout -k 10 500 python bench.py --gpus $n $X $F --crowded > gpurun_out/reh_crowded_$n.json 2> gpurun_out/reh_crowded_$n.err
  echo done $n
done
