"""GPU box: the *_stamp timing builds of scripts/exp/split_variants.sh -- shader cycles and clock of the last hidden layer's k loop
(79 slices of 96 MFMAs per wave), per wave, from s_memtime / s_memrealtime inside the kernel."""
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
net = dlc.SDAV(seed=1, dtype="f16x2", weight_scale=sys.argv[2])
net.transform_tensor(x[:2])
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:                     # (the clock settles under load)
    h = net.transform_tensor(x)
torch.cuda.synchronize()
ws = [t for k, t in eng._ws.items() if k[0] == "sdav_split"][0]
mp, M = 32000, 31890
piece0 = ws[:mp * 80 * 64].view(torch.int64).reshape(80, mp, 8)          # [slice][row][64 bytes]
d = piece0[:23, M:M + 110].reshape(-1, 2)[:125 * 10 * 8].cpu().numpy()     # 4 entries of 16 bytes per row, 110 rows per slice
if "epistamp" in sys.argv[1]:
    e = piece0[23:46, M:M + 110].reshape(-1, 2)[:125 * 10 * 8].cpu().numpy()[:, 0].astype(np.float64)
    c0 = d[:, 0].astype(np.float64)
    print("%%-30s %%-9s k loop %%.0f cycles per wave (median), loop end -> kernel end %%.0f cycles (median; max %%.0f) = %%.1f %%%% of the loop"
          %% (os.path.basename(sys.argv[1]), sys.argv[2], np.median(c0[c0 > 0]), np.median(e[e > 0]), e.max(), 100 * np.median(e[e > 0]) / np.median(c0[c0 > 0])), flush=True)
    sys.exit(0)
cyc, ticks = d[:, 0].astype(np.float64), d[:, 1].astype(np.float64)
ok = ticks > 0
if "barstamp" in sys.argv[1]:
    w = np.arange(len(cyc)) %% 8
    print("%%-34s %%-9s loop cycles %%.0f per slice; in the barrier per slice: all waves %%.0f, loaders (0-3) %%.0f, others (4-7) %%.0f; by wave %%s"
          %% (os.path.basename(sys.argv[1]), sys.argv[2], np.median(cyc[ok]) / 79, np.median(ticks[ok]) / 79, np.median(ticks[ok & (w < 4)]) / 79,
             np.median(ticks[ok & (w >= 4)]) / 79, [int(np.median(ticks[ok & (w == k)]) / 79) for k in range(8)]), flush=True)
    sys.exit(0)
print("%%-28s %%-9s loop cycles per wave: median %%.0f (per slice %%.0f; the matrix pipe alone: 3072), clock %%.3f GHz, loop %%.1f us"
      %% (os.path.basename(sys.argv[1]), sys.argv[2], np.median(cyc[ok]), np.median(cyc[ok]) / 79, np.median(cyc[ok] / ticks[ok]) * 0.1,
         np.median(ticks[ok]) / 100), flush=True)
''' % R
for v in sys.argv[1:] or ["stamp", "bare_stamp", "nodma_stamp", "nolds_stamp"]:
    for scale in ("reference", "fan_in"):
        lib = os.path.join(R, "var_build", "lib_split_%s.so" % v)
        if os.path.exists(lib):
            subprocess.run([sys.executable, "-c", code, lib, scale])
