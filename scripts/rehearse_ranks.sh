# GPU box only: bench.py at the full size with 1, 2 and 4 ranks (the N > 1 runs: all ranks on the box's ONE GPU over gloo),
# plain and --crowded; the one-line summaries go to gpurun_out/rehearsal.txt (copy to profiles/<round>_multirank_rehearsal.txt).
set -e
F="--steps 20 --warmup 5 --no-paths --no-cpu-baseline --no-power-probe --no-shard-emulation --no-configs --no-rccl-smoke"
for n in 1 2 4; do
  if [ $n = 1 ]; then X=""; else X="--backend gloo --share-gpu"; fi
  timeout -k 10 500 python bench.py --gpus $n $X $F --detail gpurun_out/reh_$n.detail.json > gpurun_out/reh_$n.json 2> gpurun_out/reh_$n.err
  timeout -k 10 500 python bench.py --gpus $n $X $F --crowded --detail gpurun_out/reh_crowded_$n.detail.json > gpurun_out/reh_crowded_$n.json 2> gpurun_out/reh_crowded_$n.err
  echo done $n
done
python3 - <<'PY' | tee gpurun_out/rehearsal.txt
import json
for kind in ("", "crowded_"):
    for n in (1, 2, 4):
        line = open("gpurun_out/reh_%s%d.json" % (kind, n)).read().strip().splitlines()
        assert len([l for l in line if l.startswith("{")]) == 1 and line[-1].startswith("{") and len(line[-1]) < 4096, (kind, n)
        d = json.load(open("gpurun_out/reh_%s%d.detail.json" % (kind, n)))
        p = d.get("pipeline") or {}
        print("%-7s n_gpus %d rows/gpu %d recall %s idx %s scores %s ms/step %.3f resolved_batches %s dropped %s line %d B smoke %s"
              % ("crowded" if kind else "plain", d["n_gpus"], d["config"]["rows_per_gpu"], d["recall_at_1"], d["topk_idx_sha256"][:16],
                 d["topk_scores_sha256"][:16], d["ms_per_step"], p.get("resolved_batches"), p.get("dropped_batches"), len(line[-1]),
                 json.dumps(d.get("rccl_smoke"))))
PY
