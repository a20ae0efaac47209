"""Micro-benchmark: time of the whole match call vs k and rows (GPU box only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
d = 4096
for n in (125_000, 1_000_000):
    db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((256, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    for k in (1, 20, 100):
        for _ in range(3):
            eng.match_topk(q, db, k)
        eng.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.match_topk(q, db, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        g = eng.profile_gemm_ms(10)
        eng.set_profiling(False)
        print("n=%d k=%d total %.3f ms gemm %.3f ms rest %.3f ms" % (n, k, dt, sum(g) / len(g), dt - sum(g) / len(g)), flush=True)
    del db
