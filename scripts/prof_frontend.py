"""Workload for a kernel trace of the patch front-end on 1063 resident frames:
rocprofv3 --kernel-trace --stats -- python3 scripts/prof_frontend.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
rgb = torch.randint(0, 256, (1063, 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
parser = dlc.CvInputParser(30, 41)
for _ in range(5):
    p = parser.parse_batch(rgb)
torch.cuda.synchronize()
print("done", p.shape)
