"""Single-frame encode latency (GPU box only): wall time per call with a sync, and the host time
to enqueue a call (no sync), for CnnVtl and SDAV at batch sizes 1, 4, 16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)

def measure(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    print("%-28s wall %7.1f us/call   host enqueue %7.1f us/call" % (name, wall * 1e6, host * 1e6), flush=True)

mode = os.environ.get("DLC_LATENCY_MODE", "0") == "1"
if mode:
    eng.set_scratch()
print("latency mode (split-K scratch):", mode)
for b in (1, 4, 16, 64, 128):
    frames = torch.randint(0, 256, (b, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
    cnn = dlc.CnnVtl(input_shape=[b, 192, 240, 3])
    measure("CnnVtl B=%d" % b, lambda: cnn.transform_tensor(frames))
    x = torch.rand((b, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
    net = dlc.SDAV(seed=1)
    measure("SDAV fp64 B=%d" % b, lambda: net.transform_tensor(x))
    net32 = dlc.SDAV(seed=1, dtype="float32")
    measure("SDAV fp32 B=%d" % b, lambda: net32.transform_tensor(x))
