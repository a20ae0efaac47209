"""Package power and shader clock while the SDAV similarity matrix (1063 x 30 x 2500) runs back to back (GPU box only):
is csrc/gram_i8.hip's int8 product kernel -- three quarters of the call -- on the board's power cap like the bf16 score GEMM?
Usage: python scripts/exp_sim_power.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import deeploopcloser_amd as dlc

eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
ds = torch.sigmoid(35.0 * torch.randn((1063, 30, 2500), generator=g, device=eng.device, dtype=torch.float64))
score = eng.distinctive_score(ds, 0.5, 0.2)
for mode in ("i8", "f64"):
    os.environ["DLC_SIM_GRAM"] = mode
    eng.sdav_similarity_matrix(ds, score, 10.0, -10.0)
    torch.cuda.synchronize()
    bench_step = lambda: eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)
    print(mode, bench.power_probe(bench_step, seconds=4.0), flush=True)
