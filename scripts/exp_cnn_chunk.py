"""CnnVtl.transform throughput vs frame_chunk (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
nf = 1063
frames = torch.randint(0, 256, (nf, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
for chunk in (213, 266, 355, 532, 1063):
    cnn = dlc.CnnVtl(input_shape=[nf, 192, 240, 3], frame_chunk=chunk)
    cnn.transform_tensor(frames); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        d8 = cnn.transform_tensor(frames)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 2
    print("frame_chunk %4d: %.1f ms for %d frames, %.0f frames/s, %.1f TF fp64" % (chunk, t * 1e3, nf, nf / t, 1.748e9 * nf / t / 1e12), flush=True)
    del cnn
    torch.cuda.empty_cache()
