"""The tolerance-mode encoder under rocm-smi: package power and shader clock while SDAV.transform(f16x2) of 1063 frames loops
(random frames, N(0,1) / fan_in weights, and all-zero weights + frames: the same instruction stream on trivial operands)."""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
n = 1063
x = torch.rand((n, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)

def probe(fn, seconds=4.0):
    samples, stop = [], threading.Event()
    def sample():
        while not stop.is_set():
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt); sc = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
            if pw and sc:
                samples.append((float(pw.group(1)), int(sc.group(1))))
            stop.wait(0.3)
    th = threading.Thread(target=sample, daemon=True); th.start()
    t0 = time.perf_counter(); calls = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); calls += 20
    dt = time.perf_counter() - t0
    stop.set(); th.join(timeout=15)
    good = samples[1:] or samples
    return dt / calls * 1e3, float(np.median([s[0] for s in good])), float(np.median([s[1] for s in good])), len(good)

for scale in ("reference", "fan_in", "zero"):
    net = dlc.SDAV(seed=1, dtype="f16x2", weight_scale="reference" if scale == "zero" else scale)
    xx = x
    if scale == "zero":
        ws, bs = net.get_weights()
        net.set_weights([w * 0 for w in ws], bs)
        xx = x * 0
    net.transform_tensor(xx[:2])
    ms, pw, clk, ns = probe(lambda: net.transform_tensor(xx))
    print("%-9s %.3f ms per call, package %.0f W, sclk %.0f MHz (%d samples)" % (scale, ms, pw, clk, ns), flush=True)
