"""Timing of the patch front-end pieces on 1063 resident frames (noise and tiled real frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd.input import CvInputParser
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N = 1063
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
rgb = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
import real_frames
sets = {"noise": rgb, "real": real_frames.tiled_bgr_frames(dlc, N).to(torch.uint8)}
parser = CvInputParser(30, 41)
for name, fr in sets.items():
    gray = eng.rgb_to_gray(fr) if hasattr(eng, "rgb_to_gray") else None
    print(name, "parse_batch %.3f ms" % timed(lambda: parser.parse_batch(fr)), flush=True)
    if gray is not None:
        print(name, "harris %.3f ms, counts mean %.1f" % (timed(lambda: eng.harris_keypoints(gray, 30)), float(eng.harris_keypoints(gray, 30)[2].float().mean())), flush=True)
gray = eng.rgb_to_gray(rgb)
for n in (1, 8, 30, 120):
    print("noise harris n=%d: %.3f ms" % (n, timed(lambda: eng.harris_keypoints(gray, n))), flush=True)
