"""Where a LoopClosureDetector.query_and_insert batch spends its 170 us (GPU box only): cProfile of the host side and
the GPU busy time from events."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine(0)
N, D, bl = 1063, 4096, 32
x = torch.randn((N, D), device=eng.device)


def stream():
    det = dlc.LoopClosureDetector(D, k=5, threshold=0.5, exclusion=30, capacity=max(64, N))
    return [det.query_and_insert(x[lo:lo + bl]) for lo in range(0, N, bl)]


stream(); torch.cuda.synchronize()
t0 = time.perf_counter(); stream(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue %.2f ms, complete %.2f ms for %d batches" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, -(-N // bl)))
pr = cProfile.Profile(); pr.enable(); stream(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
