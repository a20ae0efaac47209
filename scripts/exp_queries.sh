#!/bin/bash
# query-frames/s against the 1M-row database for several batch sizes (GPU box only)
for q in 4 8 64 128 192 256 300; do
  timeout -k 10 300 python bench.py --queries $q --steps 40 --warmup 5 --no-cpu-baseline --no-power-probe 2>/dev/null > /tmp/b.json || exit 1
  python - "$q" <<'PY'
import sys, json
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("queries", sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4),
      "gemm ms", round(d["roofline"]["kernel_ms"], 4), "recall", d["recall_at_1"])
PY
done
