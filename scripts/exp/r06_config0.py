"""configs[0] end to end (20 real frames -> patches -> SDAV -> 20 x 20 cosine matrix, host to host) with and without the
engine's latency mode (split-K for the 600-row fp64 GEMMs), stage by stage."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import pipeline
import real_frames, config1_common as c1
eng = dlc.default_engine()
frames = real_frames.tiled_u8_frames(dlc, 20)
parser = dlc.CvInputParser(30, 41)
net = dlc.SDAV(seed=c1.SEED, weight_scale="fan_in")
def config0():
    d_ = pipeline.sdav_descriptors_from_frames(frames, net, parser)
    st = eng.normalize(d_.view(20, -1), "bf16", center=True)
    return eng.download(eng.cosine_scores(st, st))
def timed(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2], r
a0, a1, ra = timed(config0)
with eng.latency_mode():
    b0, b1, rb = timed(config0)
print("configs[0] host to host: one-pass GEMMs %.3f ms (median %.3f); latency mode %.3f ms (median %.3f); max |diff| of the matrices %.3g"
      % (a0, a1, b0, b1, float(np.abs(ra - rb).max())))
x = torch.rand((20, 30, 1681), dtype=torch.float64, device=eng.device)
def enc(): return net.transform_tensor(x)
def tdev(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
e_a = tdev(enc)
with eng.latency_mode():
    e_b = tdev(enc)
print("SDAV.transform_tensor of 20 frames: %.3f ms, latency mode %.3f ms" % (e_a, e_b))
