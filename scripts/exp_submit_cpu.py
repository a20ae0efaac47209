"""CPU cost of one MatchPipeline.submit() / one-shot match_topk call (GPU box only): a database small enough that the
GPU is never the bottleneck, cProfile over 2000 steps."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc

eng = dlc.default_engine(0)
n, d, nq, k = 8192, 4096, 256, 20
db = dlc.KeyframeDatabase(torch.randn((n, d), device=eng.device), dtype="bf16")
q = eng.normalize(torch.randn((nq, d), device=eng.device), "bf16")
pipe = dlc.MatchPipeline(db, k, depth=2)
for _ in range(50):
    t = pipe.submit(q)
pipe.drain()
torch.cuda.synchronize()


def loop(steps):
    for _ in range(steps):
        pipe.submit(q)


t0 = time.perf_counter(); loop(2000); t1 = time.perf_counter()
pipe.drain(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("submit: %.1f us per step issued, %.1f us per step to completion" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
pr = cProfile.Profile(); pr.enable(); loop(2000); pr.disable(); pipe.drain()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
out = (torch.empty((nq, k), dtype=torch.float32, device=eng.device), torch.empty((nq, k), dtype=torch.int64, device=eng.device))
t0 = time.perf_counter()
for _ in range(2000):
    db.match_topk(q, k, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("one-shot: %.1f us per step issued, %.1f us to completion" % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
