"""SDAV training-step time (GPU box only): reference batch of 10 frames, each layer, one-pass vs
latency mode (split-K scratch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 1:                      # another build of the library (A/B)
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((10, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
for mode in (False, True):
    eng.set_scratch(dlc.engine.SCRATCH_BYTES if mode else 0)
    net = dlc.SDAV(seed=1)
    for layer in range(5):
        masks = [net._mask(l) for l in range(layer + 1)]
        for _ in range(3):
            net.train_step(layer, x, masks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            net.train_step(layer, x, masks)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            net.train_step(layer, x, masks)
        host = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        print("latency mode %-5s layer %d: %.2f ms/step  (host enqueue %.2f ms)" % (mode, layer, wall * 1e3, host * 1e3), flush=True)
eng.set_scratch(0)
