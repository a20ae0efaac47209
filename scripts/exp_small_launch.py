"""Small launches of the LDS-DMA fp64 GEMM: time vs tile count (GPU box only).  argv[1]: library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
if len(sys.argv) > 1:
    L._lib = None
    L.LIB_PATH = os.path.abspath(sys.argv[1])
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


K = 2304
for M, N in ((16640, 128), (16640, 256), (16640, 384), (16640, 512), (8320, 384), (4096, 384), (32768, 384), (65536, 128), (65536, 384)):
    a = torch.rand((M, K), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((K, N), generator=g, device=eng.device, dtype=torch.float64)
    ms = timed(lambda: eng.gemm_bias_act(a, b, None))
    tiles = -(-M // 256) * -(-N // 128)
    print("M %6d N %4d: %4d tiles  %.3f ms  %.1f TF" % (M, N, tiles, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
