"""Where the 0.1 ms behind the score GEMM goes (GPU box only): the fused selection (dlc_cosine_select_topk), the group
selection alone, the re-score alone, at the benchmark's 1 M x 4096 / 256 queries / k = 20 and at one of 8 shards.
Library: argv[1] (default: the shipped one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

if len(sys.argv) > 1:
    L._lib = None
    L.LIB_PATH = os.path.abspath(sys.argv[1])
eng = dlc.default_engine(0)
d, nq, k = 4096, 256, 20
kg = eng.groups_per_query(k)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (1_000_000, 125_000):
    db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((nq, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
    sc = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
    ix = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
    gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
    gm = torch.empty((nq, kg), dtype=torch.float32, device=eng.device)
    eng.score_groups(q, db, k, ws)
    t_gemm = timed(lambda: eng.score_groups(q, db, k, ws), reps=10)
    t_fused = timed(lambda: eng.select_topk(q, db, k, ws, sc, ix))
    t_sel = timed(lambda: eng.select_groups(q, db, k, ws, gi, gm))
    t_res = timed(lambda: eng.rescore_topk(q, db, k, gi, gm, sc, ix))
    t_all = timed(lambda: eng.match_topk(q, db, k), reps=10)
    print("rows %8d  score GEMM %7.1f us  fused selection %6.1f us  (groups only %6.1f, re-score only %6.1f)  match_topk %7.1f us"
          % (n, t_gemm, t_fused, t_sel, t_res, t_all), flush=True)
    del db, ws
