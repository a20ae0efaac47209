"""GPU box: time the SDAV similarity's gram_i8_kernel in the exp_build/lib_gram_*.so variants (scripts/exp/gram_variants.sh),
each in its own child process (one library per process).  Kernel time = the library's own HIP-event pair."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import deeploopcloser_amd._lib as L
L.LIB_PATH = sys.argv[1]
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
ds = torch.sigmoid(35.0 * torch.randn((1063, 30, 2500), generator=g, device=eng.device, dtype=torch.float64))
score = eng.distinctive_score(ds, 0.5, 0.2)
best = 1e9
for rep in range(6):
    eng.set_profiling(True)
    eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False, no_host_sync=True)
    torch.cuda.synchronize()
    ms = eng.profile_gemm_ms(8)
    eng.set_profiling(False)
    best = min(best, sum(ms))
print("%%-40s gram_i8_kernel %%.3f ms" %% (os.path.basename(sys.argv[1]), best), flush=True)
''' % ROOT
for lib in sorted(os.listdir(os.path.join(ROOT, "exp_build"))):
    if lib.startswith("lib_gram_") and lib.endswith(".so"):
        subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, "exp_build", lib)], check=False)
