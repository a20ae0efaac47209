"""LoopClosureDetector.query_and_insert over 1063 frames ONE frame at a time (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N, D, k, excl = 1063, 4096, 5, 30
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
def stream():
    det = dlc.LoopClosureDetector(D, k=k, threshold=0.5, exclusion=excl, capacity=max(64, N))
    return [det.query_and_insert(xs[lo:lo + 1]) for lo in range(N)]
stream(); torch.cuda.synchronize(); t0 = time.perf_counter()
R = 5
for _ in range(R): stream()
torch.cuda.synchronize(); print("one frame at a time: %.3f ms per pass, %.1f us per frame" % ((time.perf_counter() - t0) / R * 1e3, (time.perf_counter() - t0) / R * 1e6 / N), flush=True)
