"""End-to-end streaming loop-closure throughput (GPU box only): synthetic 192x240 frames ->
CnnVtl int8 descriptors -> LoopClosureDetector.query_and_insert, in batches of B frames."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
T = int(os.environ.get("DLC_FRAMES", "2048"))
period = 700                                         # the trajectory revisits each place after 700 frames
places = torch.randint(0, 256, (period, 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
for B in (1, 16, 128):
    eng.set_scratch(dlc.engine.SCRATCH_BYTES if B <= 16 else 0)      # latency mode for small batches
    cnn = dlc.CnnVtl(input_shape=[B, 192, 240, 3])
    det = None
    found = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for lo in range(0, T, B):
        ids = torch.arange(lo, min(lo + B, T), device=eng.device) % period
        noise = torch.randint(0, 8, (ids.numel(), 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
        frames = (places[ids] // 2 + noise).to(torch.float64)
        desc = cnn.transform_tensor(frames).to(torch.float32)
        if det is None:
            det = dlc.LoopClosureDetector(desc.shape[1], k=5, threshold=0.5, exclusion=50, center=True, capacity=4096)
        s, i = det.query_and_insert(desc)
        fid = torch.arange(lo, lo + ids.numel(), device=eng.device)
        revisit = fid >= period + 50                      # an earlier visit of the place is old enough to be matched
        found += int((revisit & (i[:, 0] >= 0) & (i[:, 0] % period == ids)).sum().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"batch": B, "latency_mode": B <= 16, "frames": T, "frames_per_s": T / dt, "ms_per_batch": dt / ((T + B - 1) // B) * 1e3,
                      "revisits_matched_top1": found, "revisits": max(0, T - period - 50)}), flush=True)


# ---- the same stream through the SDAV path: patches (GPU front-end, Harris key-points) -> SDAV fp64 ->
# flattened 75 000-d place descriptor -> detector
T2 = 512
parser = dlc.CvInputParser(30, 41)
net = dlc.SDAV(seed=1)
for B in (1, 16):
    eng.set_scratch(dlc.engine.SCRATCH_BYTES)
    det = None
    found = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for lo in range(0, T2, B):
        ids = torch.arange(lo, min(lo + B, T2), device=eng.device) % 200
        noise = torch.randint(0, 8, (ids.numel(), 192, 240, 3), generator=g, device=eng.device, dtype=torch.uint8)
        rgb = places[ids] // 2 + noise
        x = parser.parse_batch(rgb)                        # grey, Harris key-points, 30 patches per frame
        h = net.transform_tensor(x)
        desc = h.reshape(ids.numel(), -1).to(torch.float32)
        if det is None:
            det = dlc.LoopClosureDetector(desc.shape[1], k=5, threshold=0.5, exclusion=20, center=True, capacity=1024)
        s, i = det.query_and_insert(desc)
        fid = torch.arange(lo, lo + ids.numel(), device=eng.device)
        found += int(((fid >= 220) & (i[:, 0] >= 0) & (i[:, 0] % 200 == ids)).sum().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"network": "sdav", "batch": B, "latency_mode": True, "frames": T2, "frames_per_s": T2 / dt,
                      "ms_per_batch": dt / ((T2 + B - 1) // B) * 1e3, "revisits_matched_top1": found,
                      "revisits": T2 - 220}), flush=True)
eng.set_scratch(0)
