"""Host-to-host latency of the drop-in NumPy surface (ndarray in, ndarray out) against the resident-tensor calls,
1063 frames (outdoor_kennedylong).  GPU box only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeploopcloser_amd as dlc

eng = dlc.default_engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1063
rng = np.random.RandomState(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


x = rng.uniform(0, 1, size=(N, 30, 1681))
net = dlc.SDAV(seed=1)
xd = torch.from_numpy(x).to(eng.device)
print("SDAV.transform_tensor (resident): %.1f ms" % timeit(lambda: net.transform_tensor(xd)))
print("upload x (429 MB) staged: %.1f ms" % timeit(lambda: eng.upload(x)))
h = net.transform_tensor(xd)
print("download h (638 MB) staged: %.1f ms" % timeit(lambda: eng.download(h)))
print("pageable torch .to(device): %.1f ms ; .cpu().numpy(): %.1f ms" % (timeit(lambda: torch.from_numpy(x).to(eng.device)), timeit(lambda: h.cpu().numpy())))
for cf in (32, 64, 128, 256, 1063):
    print("SDAV.transform(ndarray) chunk_frames=%d: %.1f ms" % (cf, timeit(lambda: net.transform(x, chunk_frames=cf))))
for th in (4, 8, 16, 32):
    eng._check(eng.lib.dlc_set_host_threads(eng.ctx, th))
    print("  host threads %d: SDAV.transform(ndarray) %.1f ms" % (th, timeit(lambda: net.transform(x))))
eng._check(eng.lib.dlc_set_host_threads(eng.ctx, 0))
del x, xd, h
f8 = rng.randint(0, 256, size=(N, 192, 240, 3)).astype(np.uint8)
cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3], seed=3, mask_seed=4)
fd = torch.from_numpy(f8).to(eng.device).to(torch.float64)
print("CnnVtl.transform_tensor (resident f64): %.1f ms" % timeit(lambda: cnn.transform_tensor(fd), reps=3))
del fd
for cf in (None, 133, 266, 355, 1063):
    print("CnnVtl.transform(uint8 ndarray) chunk_frames=%s: %.1f ms" % (cf, timeit(lambda: cnn.transform(f8, chunk_frames=cf), reps=3)))
f64 = f8.astype(np.float64)
for cf in (None, 133, 266, 355):
    print("CnnVtl.transform(float64 ndarray) chunk_frames=%s: %.1f ms" % (cf, timeit(lambda: cnn.transform(f64, chunk_frames=cf), reps=3)))
