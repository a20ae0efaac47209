"""Is the implicit-im2col loader or the GEMM shape what holds the conv layers at ~45 TF?  Times a
plain fp64 GEMM of each conv layer's (M, N, K) for 128 frames next to the conv call (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
if len(sys.argv) > 1:                      # another build of the library
    L._lib = None
    L.LIB_PATH = os.path.abspath(sys.argv[1])
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
F = int(os.environ.get('DLC_FRAMES', '128'))
layers = [("conv2", 22, 28, 96, 5, 256, 2), ("conv3", 10, 13, 256, 3, 384, 1), ("conv4", 10, 13, 384, 3, 384, 1), ("conv5", 10, 13, 384, 3, 256, 1)]
for name, h, w, c, ks, cout, pad in layers:
    M, K, N = F * h * w, ks * ks * c, cout
    a = torch.rand((M, K), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((K, N), generator=g, device=eng.device, dtype=torch.float64)
    bias = torch.rand((N,), generator=g, device=eng.device, dtype=torch.float64)
    tp = t(lambda: eng.gemm_bias_act(a, b, bias, act=L.DLC_ACT_RELU))
    x = torch.rand((F, h, w, c), generator=g, device=eng.device, dtype=torch.float64)
    tc = t(lambda: eng.conv2d(x, b, bias, ks, ks, 1, pad, pad, h, w, L.DLC_ACT_RELU))
    fl = 2.0 * M * N * K
    print("%s M=%d N=%d K=%d: plain GEMM %.3f ms (%.1f TF)   implicit conv %.3f ms (%.1f TF)" %
          (name, M, N, K, tp * 1e3, fl / tp / 1e12, tc * 1e3, fl / tc / 1e12), flush=True)
    del a, b, x
