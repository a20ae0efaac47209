"""SDAV.transform of 1063 frames in the tolerance mode (f16x2), for rocprofv3 / timing: per call the wall time (stream events)
and the five gemm_split_f16_kernel launches (the library's HIP events around each).   usage: prof_sdav_split.py [frames] [weight scale] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1063
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
x = torch.rand((n, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1, dtype="f16x2", weight_scale=sys.argv[2] if len(sys.argv) > 2 else "reference")
net.transform_tensor(x[:2])
calls, layers = [], []
for rep in range(reps):
    eng.set_profiling(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    h = net.transform_tensor(x)
    e1.record()
    torch.cuda.synchronize()
    k = eng.profile_gemm_ms(16)
    eng.set_profiling(False)
    calls.append(e0.elapsed_time(e1)); layers.append(k)
layers = np.array(layers)
calls = np.array(calls)
flops = 3 * 2.0 * 30 * n * (1681 * 2500 + 4 * 2500 * 2500)
print("f16x2 %d frames, %d calls: call ms min %.3f median %.3f (first %.3f); layers ms (median) %s sum %.3f; %.3f PF at the median call = %.3f of 2.5"
      % (n, reps, calls.min(), np.median(calls), calls[0], np.round(np.median(layers, axis=0), 3).tolist(), np.median(layers, axis=0).sum(),
         flops / np.median(calls) / 1e12, flops / np.median(calls) / 1e12 / 2.5), flush=True)
