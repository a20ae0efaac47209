"""GPU box: time dlc_sdav_distinctive_score (1063 x 30 x 2500, with the column extremes) with each variant library of
scripts/exp/ds_variants.sh, one child process per library."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import sys, time, torch
sys.path.insert(0, %r)
import deeploopcloser_amd._lib as L
L.LIB_PATH = sys.argv[1]
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
ds = torch.sigmoid(35.0 * torch.randn((1063, 30, 2500), generator=g, device=eng.device, dtype=torch.float64))
ref = None
for _ in range(3): s, r = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): s, r = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)
torch.cuda.synchronize()
print("%%-28s %%.3f ms  checksum %%r" %% (sys.argv[1].split("/")[-1], (time.perf_counter() - t0) / 20 * 1e3, float(s.double().sum())), flush=True)
''' % root
for lib in sorted(glob.glob(os.path.join(root, "deeploopcloser_amd", "libdlc_ds_*.so"))):
    subprocess.run([sys.executable, "-c", child, lib], check=False)
