"""GPU box: distinctive_score_kernel at the reference's size, with and without the range tracking (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
ds = torch.rand((1063, 30, 2500), generator=g, device=eng.device, dtype=torch.float64)
for with_range in (False, True):
    for _ in range(3):
        eng.distinctive_score(ds, 0.5, 0.2, with_range=with_range)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.distinctive_score(ds, 0.5, 0.2, with_range=with_range)
    e1.record(); torch.cuda.synchronize()
    print("distinctive_score with_range=%s: %.3f ms" % (with_range, e0.elapsed_time(e1) / 20), flush=True)
