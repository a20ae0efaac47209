"""Where the selection's time goes at the benchmark size (GPU box only): fused call vs its two
halves (group selection; re-score + final top-k), non-cooperative (512-thread) forms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
n, d, nq, k = 1_000_000, 4096, 256, 20
g = torch.Generator(device=eng.device); g.manual_seed(0)
db = torch.randn((n, d), generator=g, device=eng.device).to(torch.bfloat16)
q = torch.randn((nq, d), generator=g, device=eng.device).to(torch.bfloat16)
ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
kg = eng.groups_per_query(k)
ids = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
mx = torch.empty((nq, kg), dtype=torch.float32, device=eng.device)
s = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
i = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
eng.score_groups(q, db, k, ws)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("fused select_topk      %.1f us" % t(lambda: eng.select_topk(q, db, k, ws, s, i)))
print("select_groups          %.1f us" % t(lambda: eng.select_groups(q, db, k, ws, ids, mx)))
print("rescore_topk           %.1f us" % t(lambda: eng.rescore_topk(q, db, k, ids, mx, s, i)))
