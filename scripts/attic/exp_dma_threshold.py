"""From how many 256 x 128 tiles on does the LDS-DMA fp64 GEMM beat the register-staged 128 x 128 kernel (GPU box only)?
SDAV.transform and CnnVtl.transform at small frame counts, shipped library vs builds with another DLC_DMA_MIN_TILES
(env DLC_EXP_LIBS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
for label, path in libs:
    L._lib = None
    L.LIB_PATH = path
    dlc.engine._default.clear()
    eng = dlc.default_engine(0)
    g = torch.Generator(device=eng.device); g.manual_seed(0)
    net = dlc.SDAV(seed=1)
    out = []
    for n in (4, 8, 16, 32, 64, 128, 256):
        x = torch.rand((n, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
        out.append("%d: %.2f" % (n, timed(lambda: net.transform_tensor(x))))
    print("%-20s SDAV.transform ms by frames  %s" % (label, "  ".join(out)), flush=True)
    out = []
    for n in (4, 8, 16, 32, 64, 128, 256):
        fr = torch.randint(0, 256, (n, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
        cnn = dlc.CnnVtl(input_shape=[n, 192, 240, 3])
        out.append("%d: %.2f" % (n, timed(lambda: cnn.transform_tensor(fr))))
    print("%-20s CnnVtl.transform ms by frames  %s" % (label, "  ".join(out)), flush=True)
    eng.close()
    dlc.engine._default.clear()
