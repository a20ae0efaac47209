"""SDAV.train_steps (graph replay) and train_step (eager) on the reference's batch of 10 frames, layers 0 and 2: ms per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
x = torch.rand((10, 30, 1681), dtype=torch.float64, device=eng.device)
for layer in (0, 2, 4):
    net = dlc.SDAV(seed=3, weight_scale="fan_in")
    with eng.latency_mode():
        net.train_steps(layer, x, 5)
        torch.cuda.synchronize()
        best = None
        for _ in range(6):
            t0 = time.perf_counter(); net.train_steps(layer, x, 20); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20 * 1e3
            best = dt if best is None or dt < best else best
        masks = [net._mask(l) for l in range(layer + 1)]
        t0 = time.perf_counter()
        for _ in range(20):
            net.train_step(layer, x, masks)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20 * 1e3
    print("layer %d: %.3f ms per step replayed (masks redrawn), %.3f eager" % (layer, best, eager), flush=True)
