"""The training step's 300-row fp64 products: one pass (LDS-DMA kernel, 64-row tiles) against the split-K form the step takes
when the context has scratch (register-staged kernel + reduce)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (m, k, n, bl) in ((300, 1681, 2500, L.DLC_B_KN), (300, 2500, 1681, L.DLC_B_NK), (300, 2500, 2500, L.DLC_B_KN), (1681, 600, 2500, L.DLC_B_KN)):
    a = torch.rand((m, k), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((k, n) if bl == L.DLC_B_KN else (n, k), generator=g, device=eng.device, dtype=torch.float64)
    one = t(lambda: eng.gemm_bias_act(a, b, blayout=bl))
    r1 = eng.gemm_bias_act(a, b, blayout=bl)
    with eng.latency_mode():
        spl = t(lambda: eng.gemm_bias_act(a, b, blayout=bl))
        r2 = eng.gemm_bias_act(a, b, blayout=bl)
    print("M=%d K=%d N=%d %s: one pass %.1f us (%.1f TF), with scratch (split-K plan) %.1f us (%.1f TF); max rel diff %.2g"
          % (m, k, n, "KN" if bl == L.DLC_B_KN else "NK", one, 2.0*m*k*n/one/1e6, spl, 2.0*m*k*n/spl/1e6, float(((r1-r2).abs()/r1.abs()).max())))
