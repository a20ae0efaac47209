"""rocprofv3 --kernel-trace target: 40 replayed layer-0 training steps (SDAV.train_steps) at the reference's batch of 10 frames."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
net = dlc.SDAV(seed=3, weight_scale="fan_in")
x = torch.rand((10, 30, 1681), dtype=torch.float64, device=eng.device)
with eng.latency_mode():
    net.train_steps(0, x, 5)
    torch.cuda.synchronize()
    net.train_steps(0, x, 40)
    torch.cuda.synchronize()
