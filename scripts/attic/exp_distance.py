"""cnn_vtl distance matrix timing for the shipped library and experimental builds (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
for label, path in libs:
    L._lib = None; L.LIB_PATH = path; dlc.engine._default.clear()
    eng = dlc.default_engine(0)
    g = torch.Generator(device=eng.device); g.manual_seed(0)
    for (n, d) in [(1063, 2239), (4000, 2243), (300, 2463)]:
        desc = torch.randint(-128, 128, (n, d), generator=g, device=eng.device, dtype=torch.int8)
        eng.cnnvtl_distance_matrix(desc); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.cnnvtl_distance_matrix(desc); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("%-22s N=%d D=%d  %.3f ms  %.2f T byte-pairs/s" % (label, n, d, best, n * n * d / best / 1e9), flush=True)
    eng.close()
