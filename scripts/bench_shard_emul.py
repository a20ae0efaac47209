"""Emulate ONE rank of an R-way sharded 1M-row database on one GPU (no collectives): per-step GPU
time of score GEMM + select groups + filtered re-score, two streams, vs the unfiltered select."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
d, nq, k, n_total = 4096, 256, 20, 1_000_000
kg = eng.groups_per_query(k)
for R in (8, 4, 2):
    n = n_total // R
    g = torch.Generator(device=eng.device); g.manual_seed(1)
    q = eng.normalize(torch.rand((nq, d), generator=g, device=eng.device), "bf16", center=True)
    shards = [eng.normalize(torch.rand((n, d), generator=g, device=eng.device), "bf16", center=True) for _ in range(R)]
    # all ranks' group maxima (computed once; in production: all-gather per batch)
    mx = []
    for sh in shards:
        ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
        gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device); gm = torch.empty((nq, kg), dtype=torch.float32, device=eng.device)
        eng.score_groups(q, sh, k, ws); eng.select_groups(q, sh, k, ws, gi, gm)
        mx.append(gm)
    all_max = torch.stack(mx)
    db = shards[0]
    s2 = torch.cuda.Stream()
    slots = [dict(ws=torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device),
                  gi=torch.empty((nq, kg), dtype=torch.int32, device=eng.device), gm=torch.empty((nq, kg), dtype=torch.float32, device=eng.device),
                  s=torch.empty((nq, k), dtype=torch.float32, device=eng.device), i=torch.empty((nq, k), dtype=torch.int64, device=eng.device),
                  e1=torch.cuda.Event(), e2=torch.cuda.Event(), busy=False) for _ in range(2)]
    for mode in ("filtered", "unfiltered", "fused-coop"):
        def step(j):
            sl = slots[j % 2]
            main = torch.cuda.current_stream()
            if sl["busy"]: main.wait_event(sl["e2"])
            eng.score_groups(q, db, k, sl["ws"], stream=main); sl["e1"].record(main)
            s2.wait_event(sl["e1"])
            with torch.cuda.stream(s2):
                if mode == "fused-coop":
                    eng.select_topk(q, db, k, sl["ws"], sl["s"], sl["i"], coop=True, stream=s2)
                else:
                    eng.select_groups(q, db, k, sl["ws"], sl["gi"], sl["gm"], coop=True, stream=s2)
                    eng.rescore_topk(q, db, k, sl["gi"], sl["gm"], sl["s"], sl["i"], all_max=all_max if mode == "filtered" else None, coop=True, stream=s2)
                sl["e2"].record(s2)
            sl["busy"] = True
        for j in range(10): step(j)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for j in range(100): step(j)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100 * 1e3
        print("R=%d rows/rank=%d %-11s %.4f ms/step -> %.0f q/s whole job (comm excluded)" % (R, n, mode, dt, nq / dt * 1e3), flush=True)
    del shards, db
