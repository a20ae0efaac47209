"""SDAV.transform / CnnVtl.transform latency by batch size, default (batch-invariant bits) vs latency mode (split-K
scratch on) (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
if len(sys.argv) > 1:                      # another build of the library
    from deeploopcloser_amd import _lib as L
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


net = dlc.SDAV(seed=1)
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    x = torch.rand((n, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
    fr = torch.randint(0, 256, (n, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
    cnn = dlc.CnnVtl(input_shape=[n, 192, 240, 3])
    a, c = timed(lambda: net.transform_tensor(x)), timed(lambda: cnn.transform_tensor(fr))
    with eng.latency_mode():
        b, d = timed(lambda: net.transform_tensor(x)), timed(lambda: cnn.transform_tensor(fr))
    print("%4d frames  SDAV.transform %.2f ms (latency mode %.2f)   CnnVtl.transform %.2f ms (latency mode %.2f)" % (n, a, b, c, d), flush=True)
