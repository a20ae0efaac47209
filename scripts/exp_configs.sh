#!/bin/bash
# one bench line per BASELINE config that fits one GPU (GPU box only)
for args in "--dtype f16" "--rows 100000" "--rows 100000 --dtype f16" "--rows 12500" "--rows 125000"; do
  timeout -k 10 300 python bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-power-probe 2>/dev/null > /tmp/b.json || exit 1
  python - "$args" <<'PY'
import sys, json
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("%-28s" % sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4),
      "gemm ms", round(d["roofline"]["kernel_ms"], 4), "GB/s", round(d["roofline"]["achieved"]), "recall", d["recall_at_1"])
PY
done
