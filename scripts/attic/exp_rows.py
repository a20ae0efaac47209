"""GEMM time vs shard rows for the shipped library and experimental builds (DLC_EXP_LIBS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
for n in (125_000, 250_000, 500_000, 1_000_000):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    db = torch.rand((n, 4096), generator=g, device="cuda").sub_(0.5).to(torch.bfloat16)
    q = torch.rand((256, 4096), generator=g, device="cuda").sub_(0.5).to(torch.bfloat16)
    for rnd in range(2):
        for label, path in libs:
            L._lib = None; L.LIB_PATH = path; dlc.engine._default.clear()
            eng = dlc.Engine(0)
            for _ in range(5): eng.match_topk(q, db, 20)
            eng.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): eng.match_topk(q, db, 20)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
            gm = sorted(eng.profile_gemm_ms(20))[10]
            eng.set_profiling(False); eng.close()
            if rnd == 1: print("n=%7d %-16s step %.4f ms  gemm %.4f ms" % (n, label, dt, gm), flush=True)
    del db
