"""Per-launch score-GEMM times of bench.py's loop (GPU box only): does the default 20-step run sit
in a slow start?  Re-implements the timed loop with the same pipeline and prints the event series."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
n, d, nq, k = 1_000_000, 4096, 256, 20
g = torch.Generator(device=eng.device); g.manual_seed(0)
rows = torch.empty((n, d), dtype=torch.bfloat16, device=eng.device)
for c0 in range(0, n, 32768):
    x = torch.rand((min(32768, n - c0), d), generator=g, device=eng.device)
    rows[c0:c0 + x.shape[0]] = eng.normalize(x, "bf16", center=True)
q = eng.normalize(torch.rand((nq, d), generator=g, device=eng.device), "bf16", center=True)
db = dlc.KeyframeDatabase(rows, dtype="bf16", stored=True)
pipe = dlc.MatchPipeline(db, k, depth=2)
torch.cuda.synchronize()
for trial, (warm, steps, idle) in enumerate([(3, 20, 0.0), (3, 20, 0.0), (3, 100, 0.0), (3, 20, 2.0), (20, 20, 0.0)]):
    time.sleep(idle)
    for _ in range(warm):
        pipe.submit(q)
    eng.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t = pipe.submit(q)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    ms = eng.profile_gemm_ms(min(steps, 256))
    eng.set_profiling(False)
    print("trial %d warmup %d steps %d idle %.0fs: %.3f ms/step; gemm first5 %s last5 %s mean %.3f" %
          (trial, warm, steps, idle, dt, np.round(ms[:5], 3).tolist(), np.round(ms[-5:], 3).tolist(), float(np.mean(ms))), flush=True)
