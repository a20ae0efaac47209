"""Per-launch duration of the headline's score GEMM over the first 250 steps after the database has been built (HIP events
of the library around each launch): does the chip need a while under THIS load before it settles?"""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import deeploopcloser_amd as dlc
import bench
eng = dlc.default_engine()
n, d, nq, k = 1_000_000, 4096, 256, 20
prng = np.random.RandomState(4321)
planted_rows = prng.choice(n, nq, replace=False)
rows, planted = bench.synth_shard(eng, n, d, 0, n, torch.bfloat16, planted_rows)
queries = eng.normalize(planted + 0.1 * torch.randn((nq, d), device=eng.device), torch.bfloat16, center=True)
db = dlc.KeyframeDatabase(rows, dtype=torch.bfloat16, stored=True)
torch.cuda.synchronize()
for idle_ms in (0, 200):
    time.sleep(idle_ms / 1e3)
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(250):
        db.match_topk(queries, k)
    torch.cuda.synchronize()
    g = np.array(eng.profile_gemm_ms(256))
    eng.set_profiling(False)
    print("after %d ms idle: GEMM ms by blocks of 25 launches: %s; wall per step %.3f" % (idle_ms, np.round(g[:250].reshape(-1, 25).mean(axis=1), 3).tolist(), (time.perf_counter() - t0) / 250 * 1e3), flush=True)
