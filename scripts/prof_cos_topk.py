"""configs[1]'s cosine half for rocprofv3: top-20 of the 1063 flattened 75 000-d SDAV place descriptors against
themselves (the bench's `cos_topk_75k` row), a few calls back to back.
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cos -- python3 scripts/prof_cos_topk.py [frames] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) > 3:                      # another build of the library
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[3])
import deeploopcloser_amd as dlc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1063
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
x = torch.rand((N, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
h = dlc.SDAV(seed=1).transform_tensor(x)
db = dlc.KeyframeDatabase(h.reshape(N, 30 * 2500), dtype="bf16", center=True)
rows = db.rows
for _ in range(3):
    top = eng.match_topk(rows, rows, 20, details=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    top = eng.match_topk(rows, rows, 20, details=True)
e1.record()
torch.cuda.synchronize()
import hashlib
sha = hashlib.sha256(top.idx.cpu().numpy().tobytes() + top.scores_f64.cpu().numpy().tobytes()).hexdigest()[:16]
print("cosine top-20, %d x %d: %.3f ms per call, %d queries through the exhaustive pass" % (N, rows.shape[1], e0.elapsed_time(e1) / calls, int((top.status == 2).sum())), "digest", sha, flush=True)
