"""The patch front-end on 1063 resident frames of 192 x 240 (noise and the tiled real frames) and the streaming cosine
detector over 1063 frames in batches of 32, a few times each -- for rocprofv3:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fe -- python3 scripts/prof_frontend.py"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd.input import CvInputParser
import real_frames
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N = 1063
def timed(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
parser = CvInputParser(30, 41)
noise = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
real = real_frames.tiled_bgr_frames(dlc, N).to(torch.uint8)
print("front-end, noise frames: %.3f ms" % timed(lambda: parser.parse_batch(noise), 10), flush=True)
print("front-end, real frames:  %.3f ms" % timed(lambda: parser.parse_batch(real), 10), flush=True)
D, k, excl, b = 4096, 5, 30, 32
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
def stream():
    det = dlc.LoopClosureDetector(D, k=k, threshold=0.5, exclusion=excl, capacity=max(64, N))
    outs = [det.query_and_insert(xs[lo:lo + b]) for lo in range(0, N, b)]
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
print("LoopClosureDetector over %d frames in batches of %d: %.3f ms" % (N, b, timed(stream, 10)), flush=True)
