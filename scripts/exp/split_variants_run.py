"""GPU box: SDAV.transform (f16x2, 1063 frames) with each timing build of scripts/exp/split_variants.sh, two interleaved rounds,
ONE process per library (results of the variants are wrong by construction: only the time is read)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import torch
import deeploopcloser_amd._lib as L
L.LIB_PATH = sys.argv[1]
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((1063, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1, dtype="f16x2")
net.transform_tensor(x[:2])
calls, layers = [], []
for rep in range(24):
    eng.set_profiling(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net.transform_tensor(x); e1.record(); torch.cuda.synchronize()
    layers.append(eng.profile_gemm_ms(16)); eng.set_profiling(False)
    calls.append(e0.elapsed_time(e1))
print("%%-30s call median %%.3f ms min %%.3f; layers %%s" %% (os.path.basename(sys.argv[1]), np.median(calls[4:]), min(calls),
      np.round(np.median(np.array(layers)[4:], axis=0), 3).tolist()), flush=True)
''' % R
libs = sys.argv[1:] or ["base", "nodma", "nobar", "nolds", "bare", "noepi", "bare_noepi"]
for rnd in range(2):
    for v in libs:
        lib = os.path.join(R, "var_build", "lib_split_%s.so" % v)
        if os.path.exists(lib):
            subprocess.run([sys.executable, "-c", code, lib])
