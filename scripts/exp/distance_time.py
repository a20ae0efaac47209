"""cnn_vtl distance matrix at 1063 and 4000 frames of 4064 int8, a few launches each (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(2)
for n, d in ((1063, 4064), (4000, 4064)):
    x = torch.randint(-128, 128, (n, d), generator=g, device=eng.device, dtype=torch.int8)
    for _ in range(3): eng.cnnvtl_distance_matrix(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.cnnvtl_distance_matrix(x)
    torch.cuda.synchronize(); print(n, d, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
