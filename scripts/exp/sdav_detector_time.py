"""SdavLoopClosureDetector.query_and_insert over 1063 frames x 30 x 2500 in batches of 32, a few passes (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
N, P, H, sb = 1063, 30, 2500, int(sys.argv[1]) if len(sys.argv) > 1 else 32
desc = torch.sigmoid(35.0 * torch.randn((N, P, H), generator=g, device=eng.device, dtype=torch.float64))
score = eng.distinctive_score(desc, 0.5, 0.2)
def stream():
    det = dlc.SdavLoopClosureDetector(score, patches=P, width=H, k=5, exclusion=30, capacity=N)
    return [det.query_and_insert(desc[lo:lo + sb]) for lo in range(0, N, sb)]
stream(); torch.cuda.synchronize()
t0 = time.perf_counter()
R = 3
for _ in range(R): stream()
torch.cuda.synchronize(); print("batches of %d: %.3f ms per pass" % (sb, (time.perf_counter() - t0) / R * 1e3), flush=True)
