#!/bin/bash
# one-GPU bench with the two-stream pipeline (--pipeline) and with one-shot calls (default), twice each (GPU box only)
for i in 1 2; do
for mode in "--pipeline" ""; do
  timeout -k 10 300 python bench.py $mode --no-cpu-baseline --no-power-probe 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
done; done
