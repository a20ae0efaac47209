"""Timings of the non-headline hot-path rows at BASELINE config-2/3 sizes (GPU box only):
SDAV encode of 1063 frames, SDAV similarity matrix 1063x1063, cnn_vtl distance matrix,
CnnVtl encode.  Prints one JSON line per path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), r

N = int(os.environ.get("DLC_FRAMES", "1063"))
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)

# --- SDAV encode, fp64 (reference arithmetic) and fp32
x = torch.rand((N, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
for dt in ("float64", "float32"):
    net = dlc.SDAV(seed=1, dtype=dt)
    t, h = timed(lambda: net.transform_tensor(x))
    flops = 2.0 * 30 * N * (1681 * 2500 + 4 * 2500 * 2500)
    print(json.dumps({"path": "SDAV.transform", "dtype": dt, "frames": N, "ms": t * 1e3, "frames_per_s": N / t,
                      "tflops": flops / t / 1e12}), flush=True)
h64 = dlc.SDAV(seed=1).transform_tensor(x).reshape(N, 30, 2500)

# --- config 2: cosine matrix + top-k over the flattened [30*2500] place descriptors, bf16
place = h64.reshape(N, 30 * 2500)
db = dlc.KeyframeDatabase(place, dtype="bf16", center=True)
qs = db.rows
t, s = timed(lambda: eng.cosine_scores(qs, db.rows))
print(json.dumps({"path": "cosine matrix (flattened SDAV descriptors)", "frames": N, "dim": place.shape[1],
                  "ms": t * 1e3, "pairs_per_s": N * N / t, "tflops": 2.0 * N * N * db.rows.shape[1] / t / 1e12}), flush=True)
t, _ = timed(lambda: eng.match_topk(qs, db.rows, 20))
print(json.dumps({"path": "cosine top-20 (flattened SDAV descriptors)", "frames": N, "ms": t * 1e3}), flush=True)
del db, qs, s, place

# --- SDAV similarity matrix (reference semantics), fp64
calc = dlc.SimilarityCalculator(h64)
t, m = timed(lambda: eng.sdav_similarity_matrix(calc._dataset_dev, calc._score, 10.0, -10.0), reps=2)
pairs = N * (N - 1) // 2
print(json.dumps({"path": "SDAV similarity matrix", "frames": N, "ms": t * 1e3, "pairs_per_s": pairs / t,
                  "gram_tflops_f64": 2.0 * (N * 30) ** 2 * 2500 / 2 / t / 1e12}), flush=True)

# --- cnn_vtl distance matrix
desc = torch.randint(-128, 128, (N, 2243), generator=g, device=eng.device, dtype=torch.int8)
t, dm = timed(lambda: eng.cnnvtl_distance_matrix(desc))
print(json.dumps({"path": "cnn_vtl distance matrix", "frames": N, "ms": t * 1e3, "pairs_per_s": N * N / t,
                  "byte_pairs_per_s": N * N * 2243 / t}), flush=True)

# --- CnnVtl encode (192x240 frames)
nf = N
frames = torch.randint(0, 256, (nf, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
cnn = dlc.CnnVtl(input_shape=[nf, 192, 240, 3])
t, d8 = timed(lambda: cnn.transform_tensor(frames), reps=2)
print(json.dumps({"path": "CnnVtl.transform", "frames": nf, "ms": t * 1e3, "frames_per_s": nf / t,
                  "tflops_f64": 1.748e9 * nf / t / 1e12}), flush=True)
