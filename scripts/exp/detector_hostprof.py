"""Host-side profile (cProfile) of LoopClosureDetector.query_and_insert over 1063 frames in batches of 32."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N, D, k, excl, b = 1063, 4096, 5, 30, 32
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
def stream():
    det = dlc.LoopClosureDetector(D, k=k, threshold=0.5, exclusion=excl, capacity=max(64, N))
    return [det.query_and_insert(xs[lo:lo + b]) for lo in range(0, N, b)]
for _ in range(3): stream()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): stream()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
