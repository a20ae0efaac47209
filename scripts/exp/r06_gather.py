"""extract_patches (CvInputParser.py:49-97 on the GPU) at 1063 frames x 30 patches of 41 x 41: ms per call; argv[1]: another build."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
if len(sys.argv) > 1:
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
gray = torch.randint(0, 256, (1063, 192, 240), generator=g, device=eng.device, dtype=torch.uint8)
kp = torch.stack([torch.randint(0, 192, (1063, 30), generator=g, device=eng.device), torch.randint(0, 240, (1063, 30), generator=g, device=eng.device)], 2).to(torch.int32)
for _ in range(3): out = eng.extract_patches(gray, kp, 41)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = eng.extract_patches(gray, kp, 41)
e1.record(); torch.cuda.synchronize()
import hashlib
print("extract_patches: %.1f us per call, digest %s" % (e0.elapsed_time(e1) / 20 * 1e3, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]))
