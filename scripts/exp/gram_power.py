"""GPU box: package power and shader clock while the SDAV similarity's gram_i8_kernel loops (rocm-smi sampled, as
bench.py's power_probe), for the shipped library or one of exp_build/lib_gram_*.so:
  python3 scripts/exp/gram_power.py [library]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
if len(sys.argv) > 1:
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import deeploopcloser_amd as dlc
import bench

eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
ds = torch.sigmoid(35.0 * torch.randn((1063, 30, 2500), generator=g, device=eng.device, dtype=torch.float64))
score, rng = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)


def step():
    eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False, no_host_sync=True, range=rng)


for _ in range(3):
    step()
torch.cuda.synchronize()
eng.set_profiling(True)
step()
torch.cuda.synchronize()
ms = sum(eng.profile_gemm_ms(8))
eng.set_profiling(False)
orig = bench.power_probe.__defaults__
p = bench.power_probe(lambda: [step() for _ in range(1)], seconds=4.0)
print("%-34s gram %.3f ms  power %s" % (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "shipped", ms, p), flush=True)
