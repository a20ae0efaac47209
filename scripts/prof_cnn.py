import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
frames = torch.randint(0, 256, (256, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
cnn = dlc.CnnVtl(input_shape=[256, 192, 240, 3])
for _ in range(3):
    d8 = cnn.transform_tensor(frames)
torch.cuda.synchronize()
