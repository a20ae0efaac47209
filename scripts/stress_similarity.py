"""GPU box: the SDAV similarity's two routes (int8 arg-min filter, fp64 Gram form) on random shapes and data kinds -- equal bit
for bit -- and the streaming form's rows against the matrix' columns.   python3 scripts/stress_similarity.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = dlc.default_engine()
rng = np.random.RandomState(seed)
g = torch.Generator(device=eng.device); g.manual_seed(seed)
bad = 0
for case in range(cases):
    p = int(rng.choice([1, 2, 5, 7, 8, 13, 16, 21, 30, 31, 32]))
    h = int(rng.choice([8, 64, 65, 129, 250, 256, 300, 512, 640, 700, 768, 1000, 1024, 2500, 2560]))
    n = int(rng.randint(2, max(3, min(400, 40000 // (p * max(1, h // 256))))))
    kind = rng.choice(["uniform", "saturated", "normal", "twins", "binary", "lowcontrast"])
    centre, dev = None, None
    if kind == "uniform":
        ds = torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
    elif kind == "normal":
        ds = 3.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64) - 1.0
    elif kind == "lowcontrast":               # columns within ~1e-3 of their own means, the means spread over [0.15, 0.88]
        centre = 0.15 + 0.73 * torch.rand((h,), generator=g, device=eng.device, dtype=torch.float64)
        ds = centre + 1e-3 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
        dev = float((ds - centre).abs().max())
    elif kind == "binary":
        ds = (torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64) < 0.5).double()
    else:
        ds = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
        if kind == "twins" and p >= 4:
            ds[:, 1] = ds[:, 0]; ds[:, 3] = ds[:, 2]
    score = eng.distinctive_score(ds, 0.5, 0.2)
    fa, ia = (t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0))
    fb, ib = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=True)
    ok = torch.equal(torch.nan_to_num(fa, posinf=1e300), torch.nan_to_num(fb, posinf=1e300)) and torch.equal(ia, ib)
    if ok and kind != "normal" and n <= 120:                     # the streaming form (values inside its fixed range [0, 1])
        kw = dict(value_range=(-1.05 * dev, 1.05 * dev), column_centre=centre) if kind == "lowcontrast" else {}
        st = dlc.SimilarityStream(score, patches=p, width=h, capacity=n, **kw)
        st.append(ds)
        f = int(rng.randint(1, n))
        row = st.query(f)
        ok = torch.equal(torch.nan_to_num(row, posinf=1e300), torch.nan_to_num(fa[:f, f], posinf=1e300))
    if not ok:
        bad += 1
    print("%3d  n=%3d p=%2d h=%4d %-9s %s" % (case, n, p, h, kind, "ok" if ok else "MISMATCH"), flush=True)
print("stress_similarity: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
