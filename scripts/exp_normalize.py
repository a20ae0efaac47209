"""l2-normalisation into the stored 16-bit form: timing at the database chunk size (32768 x 4096 fp32) and at the
flattened-SDAV-descriptor size (1063 x 75008 fp64) (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
for (n, d, dt) in ((32768, 4096, torch.float32), (1063, 75008, torch.float64), (1063, 75000, torch.float64), (4000, 20000, torch.float32)):
    x = torch.rand((n, d), generator=g, device=eng.device, dtype=dt)
    for center in (False, True):
        eng.normalize(x, "bf16", center=center); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = eng.normalize(x, "bf16", center=center); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        nbytes = x.numel() * x.element_size() + y.numel() * 2
        print("%6d x %6d %s center=%d: %.3f ms  %.2f TB/s (one read + one write)" % (n, d, str(dt)[6:], center, best, nbytes / best / 1e9), flush=True)
