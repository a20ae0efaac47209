#!/bin/bash
# Sample rocm-smi while the bench loops (GPU box only).
cd "$(dirname "$0")/.."
python bench.py --steps 2000 --warmup 3 --no-cpu-baseline > /tmp/bench_long.log 2>&1 &
BP=$!
sleep 6
for i in 1 2 3; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | head -12
  echo ---
  sleep 1
done
wait $BP
tail -1 /tmp/bench_long.log | cut -c1-200
