"""GPU box: SDAV.transform (f16x2, 1063 frames) with each timing build of scripts/exp/split_variants.sh, interleaved rounds in
ONE process per library (results of the variants are wrong by construction: only the time is read)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import deeploopcloser_amd._lib as L
L.LIB_PATH = sys.argv[1]
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((1063, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1, dtype="f16x2")
net.transform_tensor(x[:2])
ts = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.transform_tensor(x); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("%%-28s min %%.2f ms  median %%.2f ms" %% (os.path.basename(sys.argv[1]), min(ts), sorted(ts)[len(ts) // 2]), flush=True)
''' % R
for rnd in range(2):
    for v in ("base", "nodma", "nobar", "nolds", "l2fed"):
        lib = os.path.join(R, "exp_build", "lib_split_%s.so" % v)
        if os.path.exists(lib):
            subprocess.run([sys.executable, "-c", code, lib])
