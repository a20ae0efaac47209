"""SDAV.transform of 1063 frames in the tolerance mode (f16x2) a few times, for rocprofv3 / timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1063
x = torch.rand((n, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1, dtype="f16x2", weight_scale=sys.argv[2] if len(sys.argv) > 2 else "reference")
net.transform_tensor(x[:2])
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = net.transform_tensor(x)
    torch.cuda.synchronize()
    print("f16x2 %d frames: %.2f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
