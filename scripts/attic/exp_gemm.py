"""GEMM experiments (GPU box only): time the score GEMM of the shipped library and of
experimental builds given as extra .so paths (env DLC_EXP_LIBS, ':'-separated)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

def run(libpath, label, n=int(os.environ.get('DLC_EXP_ROWS', '1000000')), d=4096, nq=256, k=20, iters=int(os.environ.get('DLC_EXP_ITERS', '10'))):
    L._lib = None
    L.LIB_PATH = libpath
    if os.environ.get("DLC_EXP_OLD_ABI") and label != "shipped":       # libraries built before ABI 2
        L.SIGNATURES.pop("dlc_cosine_scores_workspace_bytes", None)
        L.SIGNATURES.pop("dlc_cosine_scores", None)
    dlc.engine._default.clear()
    eng = dlc.Engine(0)
    db = torch.randn((n, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    q = torch.randn((nq, d), device=eng.device, dtype=torch.float32).to(torch.bfloat16)
    for _ in range(3):
        eng.match_topk(q, db, k)
    eng.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.match_topk(q, db, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters * 1e3
    g = eng.profile_gemm_ms(iters)
    gm = sorted(g)[len(g) // 2]
    print("%-28s total %.3f ms  gemm median %.3f ms (min %.3f)  %.0f TF  %.2f TB/s" %
          (label, dt, gm, min(g), 2.0 * nq * n * d / gm / 1e9, n * d * 2 / gm / 1e9), flush=True)
    del db, q
    eng.close()

base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeploopcloser_amd", "libdlc_hip.so")
libs = [("shipped", base)] + [(os.path.basename(p), p) for p in os.environ.get("DLC_EXP_LIBS", "").split(":") if p]
if os.environ.get("DLC_EXP_SKIP_SHIPPED"):
    libs = libs[1:]
for rnd in range(int(os.environ.get("DLC_EXP_ROUNDS", "2"))):
    for label, path in libs:
        run(path, label)
