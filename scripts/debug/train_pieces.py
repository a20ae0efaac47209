import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
net = dlc.SDAV(seed=3)
x = torch.rand((10, 30, 1681), dtype=torch.float64, device=eng.device)
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with eng.latency_mode():
    net.train_steps(0, x, 5)
    g = net._step_graphs[(0, 10)]
    print("graph replay alone: %.3f ms" % t(lambda: g["graph"].replay()))
    print("mask fill alone:    %.3f ms" % t(lambda: net._fill_mask(g["masks"][0], 0)))
    masks = [net._mask(0)]
    print("eager step (fixed masks): %.3f ms" % t(lambda: net.train_step(0, x, masks)))
    print("train_steps per step: %.3f ms" % (t(lambda: net.train_steps(0, x, 50), n=2) / 50))
