"""GPU box: time plain fp64 GEMM shapes with the library given as argv[1] (scripts/exp/dma_tile_variants.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if len(sys.argv) > 1:
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
for m, n, k in ((1681, 2500, 600), (2500, 2500, 600), (1681, 2500, 300), (3000, 2500, 1681), (600, 2500, 2500)):
    a = torch.rand((m, k), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((k, n), generator=g, device=eng.device, dtype=torch.float64)
    for _ in range(5): c = eng.gemm_bias_act(a, b, None, act=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): c = eng.gemm_bias_act(a, b, None, act=0)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 50 * 1e6
    print("%s  %d x %d x %d: %.1f us, %.1f TF" % (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "shipped", m, n, k, us, 2.0 * m * n * k / us / 1e6), flush=True)
