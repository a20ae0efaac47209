"""LoopClosureDetector.query_and_insert over 1063 frames in batches of 32: wall time per pass (the bench row), for
rocprofv3 --kernel-trace --stats to split it into GPU time and host time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N, D, k, excl, b = 1063, 4096, 5, 30, 32
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
def stream():
    det = dlc.LoopClosureDetector(D, k=k, threshold=0.5, exclusion=excl, capacity=max(64, N))
    outs = [det.query_and_insert(xs[lo:lo + b]) for lo in range(0, N, b)]
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
for _ in range(3): stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 20
for _ in range(R): stream()
torch.cuda.synchronize(); print("stream: %.3f ms per pass (%d passes)" % ((time.perf_counter() - t0) / R * 1e3, R + 3), flush=True)
