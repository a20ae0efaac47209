import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((1063, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1)
for _ in range(3):
    h = net.transform_tensor(x)
torch.cuda.synchronize()
