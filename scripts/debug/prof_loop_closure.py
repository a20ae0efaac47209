"""GPU box: the streaming loop-closure query as bench.py times it (1063 frames of 4096-d in batches of 32), for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
N, D, bl = 1063, 4096, 32
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    det = dlc.LoopClosureDetector(D, k=5, threshold=0.5, exclusion=30, capacity=max(64, N))
    outs = [det.query_and_insert(xs[lo:lo + bl]) for lo in range(0, N, bl)]
    torch.cuda.synchronize()
    print("loop closure, %d batches: %.2f ms" % (len(outs), (time.perf_counter() - t0) * 1e3), flush=True)
