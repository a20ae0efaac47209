"""Host-to-host timings of the reference-surface calls (numpy in, numpy out): what the PCIe transfers add to the
resident-input figures bench.py reports (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

N = 1063
rng = np.random.RandomState(0)
x = rng.uniform(0, 1, size=(N, 30, 1681))
net = dlc.SDAV(seed=1)
net.transform(x[:8])
for _ in range(2):
    t0 = time.perf_counter(); h = net.transform(x); t1 = time.perf_counter()
    print("SDAV.transform numpy->numpy %d frames: %.1f ms (%.0f frames/s); in %.0f MB, out %.0f MB" %
          (N, (t1 - t0) * 1e3, N / (t1 - t0), x.nbytes / 1e6, h.nbytes / 1e6), flush=True)
frames = rng.randint(0, 256, size=(N, 192, 240, 3)).astype(np.uint8)
cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3])
cnn.transform(frames[:8])
for _ in range(2):
    t0 = time.perf_counter(); d = cnn.transform(frames); t1 = time.perf_counter()
    print("CnnVtl.transform uint8 numpy->numpy %d frames: %.1f ms (%.0f frames/s); in %.0f MB, out %.1f MB" %
          (N, (t1 - t0) * 1e3, N / (t1 - t0), frames.nbytes / 1e6, d.nbytes / 1e6), flush=True)
q = rng.standard_normal((256, 4096)).astype(np.float32)
db = dlc.KeyframeDatabase(rng.standard_normal((100000, 4096)).astype(np.float32), dtype="bf16")
db.match_topk(q, 20)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); s, i = db.match_topk(q, 20); s = s.cpu().numpy(); i = i.cpu().numpy(); t1 = time.perf_counter()
    print("match_topk 256 fp32 host queries vs resident 100k x 4096: %.2f ms host to host" % ((t1 - t0) * 1e3), flush=True)
