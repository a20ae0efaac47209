"""Workload for the PMC passes of the dense fp64 GEMM (scripts/collect_gemm_pmc.sh): the three callers of
gemm_bias_act_kernel<double> at BASELINE configs[1] / configs[2] size -- SDAV.transform (1063 frames),
CnnVtl.transform (1063 frames of 192x240), the Gram blocks of the SDAV similarity matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc

N = int(os.environ.get("DLC_FRAMES", "1063"))
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((N, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1)
for _ in range(2):
    h = net.transform_tensor(x)
torch.cuda.synchronize()
desc = h.reshape(N, 30, 2500)
score = eng.distinctive_score(desc, 0.5, 0.2)
for _ in range(2):
    eng.sdav_similarity_matrix(desc, score, 10.0, -10.0)
torch.cuda.synchronize()
del x, h, desc
frames = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3])
for _ in range(2):
    d8 = cnn.transform_tensor(frames)
torch.cuda.synchronize()
print("done", d8.shape)
