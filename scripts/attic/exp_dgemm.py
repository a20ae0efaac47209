"""Dense fp64 GEMM experiments (GPU box only): time the three callers of gemm_bias_act_kernel<double> at
configs[1] / configs[2] size for the shipped library and for experimental builds (env DLC_EXP_LIBS, ':'-separated)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

N = int(os.environ.get("DLC_FRAMES", "1063"))


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def run(libpath, label):
    L._lib = None
    L.LIB_PATH = libpath
    dlc.engine._default.clear()
    eng = dlc.default_engine(0)
    g = torch.Generator(device=eng.device); g.manual_seed(0)
    x = torch.rand((N, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
    net = dlc.SDAV(seed=1)
    t_sdav = timed(lambda: net.transform_tensor(x))
    h = net.transform_tensor(x).reshape(N, 30, 2500)
    score = eng.distinctive_score(h, 0.5, 0.2)
    t_sim = timed(lambda: eng.sdav_similarity_matrix(h, score, 10.0, -10.0), reps=2)
    del x, h
    frames = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
    cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3])
    t_cnn = timed(lambda: cnn.transform_tensor(frames), reps=2)
    del frames
    print("%-28s SDAV.transform %.2f ms (%.1f TF)  similarity %.2f ms  CnnVtl.transform %.2f ms (%.1f TF)" %
          (label, t_sdav, 1.752e9 * N / t_sdav / 1e9, t_sim, t_cnn, 1.748e9 * N / t_cnn / 1e9), flush=True)
    eng.close()
    dlc.engine._default.clear()
    torch.cuda.empty_cache()


base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
for rnd in range(int(os.environ.get("DLC_EXP_ROUNDS", "2"))):
    for label, path in libs:
        run(path, label)
