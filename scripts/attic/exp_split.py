"""Split-K heuristic check (GPU box only): match_topk time for small databases with the shipped
library and with an experimental build that never splits (DLC_EXP_LIBS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
shapes = [(256, 1063, 4096), (256, 4096, 4096), (256, 12500, 4096), (256, 25000, 4096), (256, 32000, 4096), (256, 50000, 4096),
          (1, 1063, 75008), (16, 5000, 75008), (256, 12500, 1024), (256, 3000, 16384)]
for label, path in libs:
    L._lib = None
    L.LIB_PATH = path
    dlc.engine._default.clear()
    eng = dlc.Engine(0)
    for nq, n, d in shapes:
        db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
        q = torch.randn((nq, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
        for _ in range(5):
            eng.match_topk(q, db, 20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.match_topk(q, db, 20)
        torch.cuda.synchronize()
        print("%-16s q=%4d n=%6d d=%6d  %.1f us" % (label, nq, n, d, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
        del db, q
    eng.close()
