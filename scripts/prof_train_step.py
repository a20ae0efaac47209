"""Workload for a kernel trace of one SDAV training step (layer 0, the reference's batch of 10 frames), as SDAV.fit
runs it (split-K scratch on):  rocprofv3 --kernel-trace --stats -- python3 scripts/prof_train_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
layer = int(os.environ.get("DLC_LAYER", "0"))
x = torch.rand((10, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=3)
masks = [net._mask(l) for l in range(layer + 1)]
with eng.latency_mode():
    for _ in range(int(os.environ.get("DLC_STEPS", "20"))):
        net.train_step(layer, x, masks)
    torch.cuda.synchronize()
print("done")
