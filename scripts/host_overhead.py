"""Host-side cost of MatchPipeline.submit vs GPU time per step (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
for n in (125_000, 1_000_000):
    g = torch.Generator(device=eng.device); g.manual_seed(1)
    rows = eng.normalize(torch.rand((n, 4096), generator=g, device=eng.device), "bf16", center=True)
    q = eng.normalize(torch.rand((256, 4096), generator=g, device=eng.device), "bf16", center=True)
    db = dlc.KeyframeDatabase(rows, stored=True)
    pipe = dlc.MatchPipeline(db, 20)
    for _ in range(10): pipe.submit(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): pipe.submit(q)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("n=%d: host submit %.1f us/step, end-to-end %.1f us/step" % (n, t_host / 200 * 1e6, t_all / 200 * 1e6), flush=True)
    # tiny problem: pure host/launch cost
    small = dlc.KeyframeDatabase(rows[:2048], stored=True)
    p2 = dlc.MatchPipeline(small, 20)
    for _ in range(10): p2.submit(q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): p2.submit(q)
    th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print("   2048-row shard: host %.1f us/step, end-to-end %.1f us/step" % (th / 200 * 1e6, ta / 200 * 1e6), flush=True)
    del rows, db, pipe
