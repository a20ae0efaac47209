"""The SDAV similarity matrix at the reference's size (1063 frames x 30 patches x 2500) a few times, for rocprofv3:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sim -- python3 scripts/prof_similarity.py [saturated|uniform|duplicates|twins|binary|real|real_fan_in] [library]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 2:                      # another build of the library
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[2])
import deeploopcloser_amd as dlc

eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(3)
n, p, h = 1063, 30, 2500
kind = sys.argv[1] if len(sys.argv) > 1 else "saturated"
if kind == "saturated":
    ds = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
elif kind == "duplicates":                 # every fifth patch a copy of another one: exact ties wherever a copy is nearest
    ds = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    flat = ds.reshape(n * p, h)
    src = torch.randint(0, n * p, (n * p // 5,), generator=g, device=eng.device)
    dst = torch.randint(0, n * p, (n * p // 5,), generator=g, device=eng.device)
    flat[dst] = flat[src].clone()
elif kind == "twins":                      # in every frame patches 1, 3, 5 are copies of 0, 2, 4: a fifth of all arg-mins are exact ties
    ds = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    ds[:, 1] = ds[:, 0]; ds[:, 3] = ds[:, 2]; ds[:, 5] = ds[:, 4]
elif kind in ("real", "real_fan_in"):      # the repo's 20 real frames tiled to 1063, N(0,1) / 1/sqrt(fan_in) weights
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import real_frames
    xs = real_frames.tiled_patches(dlc, n)
    ds = dlc.SDAV(seed=4, weight_scale="fan_in" if kind == "real_fan_in" else "reference").transform_tensor(xs).reshape(n, p, h)
elif kind == "binary":                     # zeros and ones only: every squared distance an integer, ties between DIFFERENT patches
    ds = (torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64) < 0.5).double()
else:
    ds = torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
# As bench.py times the row: warm-up calls, then calls back to back (a synchronise between two calls lets the chip idle
# down and the first kernels behind it run at the clock it dropped to -- the r05 profile's 6.8 ms average of FOUR calls
# with a synchronise around each was that, beside 5.6-6.3 ms from HIP events in a loop).  rocprofv3's per-kernel average
# is over warm-up and timed calls alike; the HIP-event figure printed here is the timed calls' only.
warm, calls = int(os.environ.get("DLC_PROF_WARM", "3")), int(os.environ.get("DLC_PROF_CALLS", "12"))
def call():
    score, rng = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)      # as SimilarityCalculator(dataset).similarity_matrix()
    stats = torch.zeros((2,), dtype=torch.int64, device=eng.device)
    eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, range=rng, stats=stats)
    return stats
for _ in range(warm):
    stats = call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    stats = call()
e1.record()
torch.cuda.synchronize()
print("%s: %.3f ms per call (HIP events, %d calls back to back after %d warm-up calls), stats %s"
      % (kind, e0.elapsed_time(e1) / calls, calls, warm, stats.tolist()), flush=True)
