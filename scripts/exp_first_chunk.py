"""Host-to-host ms of the end-to-end pipelines (1063 tiled real frames) by the size of the first upload chunk."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import deeploopcloser_amd as dlc
from deeploopcloser_amd import pipeline
import real_frames
eng = dlc.default_engine()
frames = real_frames.tiled_u8_frames(dlc, 1063)
bgr = np.ascontiguousarray(frames[..., ::-1])
parser = dlc.CvInputParser(30, 41)
nets = {"fp64": dlc.SDAV(seed=4, weight_scale="fan_in"), "f16x2": dlc.SDAV(seed=4, weight_scale="fan_in", dtype="f16x2")}
cnn = dlc.CnnVtl(input_shape=[1063, 192, 240, 3], seed=3, mask_seed=4)
def best(fn, reps=4):
    fn(); b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
for first in (None, 16, 32, 48, 64, 96, 128):
    pipeline.FIRST_CHUNK_FRAMES = first
    print("first chunk %s: sdav fp64 %.2f ms, f16x2 %.2f ms, cnn_vtl %.2f ms" % (first,
          best(lambda: pipeline.sdav_similarity_matrix_from_frames(frames, nets["fp64"], parser)),
          best(lambda: pipeline.sdav_similarity_matrix_from_frames(frames, nets["f16x2"], parser)),
          best(lambda: pipeline.cnn_vtl_distance_matrix_from_frames(bgr, cnn))), flush=True)
