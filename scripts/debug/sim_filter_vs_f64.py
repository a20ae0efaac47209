"""GPU box: the int8 arg-min filter against the fp64 Gram form on random descriptors; prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if len(sys.argv) > 1:
    import deeploopcloser_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
for (n, p, h) in [(3000, 30, 256), (1500, 8, 192), (2, 1, 64), (2, 32, 64), (5, 32, 4096)]:
    g = torch.Generator(device=eng.device); g.manual_seed(n)
    ds = torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
    score = eng.distinctive_score(ds, 0.5, 0.2)
    a, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)
    b, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False, force_f64=True)
    bad = (a != b).nonzero()
    print((n, p, h), "differ:", bad.shape[0], flush=True)
    if bad.shape[0]:
        import collections
        rows = collections.Counter(bad[:, 0].tolist()); cols = collections.Counter(bad[:, 1].tolist())
        pass
