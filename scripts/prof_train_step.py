"""rocprofv3 target: 60 layer-0 training steps at the reference's batch of 10 frames (SDAV.train_steps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
net = dlc.SDAV(seed=3)
x = torch.rand((10, 30, 1681), dtype=torch.float64, device=eng.device)
masks = [net._mask(0)]
with eng.latency_mode():
    for _ in range(60):
        net.train_step(0, x, masks)
torch.cuda.synchronize()
