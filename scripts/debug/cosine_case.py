import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import deeploopcloser_amd as dlc
from oracle import cosine as ocos
eng = dlc.default_engine()
n, d, q, k = 1226, 30, 5, 35
for seed in range(6):
    for crowd in (False, True):
        rng = np.random.RandomState(seed)
        g = torch.Generator(device=eng.device); g.manual_seed(seed)
        x = torch.randn((n, d), generator=g, device=eng.device)
        db = eng.normalize(x, "bf16")
        if crowd:
            bits = db.view(torch.int16)
            where = rng.choice(n, size=200, replace=False)
            for j, r in enumerate(where.tolist()):
                bits[r] = bits[0]
                bits[r, (13 * j) % db.shape[1]] += 1 if j % 2 else -1
        qs = eng.normalize(torch.randn((q, d), generator=g, device=eng.device), "bf16")
        qs[0] = db[0]
        top = eng.match_topk(qs, db, k, details=True)
        es, ei = ocos.cosine_topk(qs.float().cpu().numpy().astype(np.float64), db.float().cpu().numpy().astype(np.float64), k)
        gi = top.idx.cpu().numpy()
        bad = np.argwhere(gi != ei)
        print("seed", seed, "crowd", crowd, "status", top.status.cpu().tolist(), "differing slots", len(bad))
        if len(bad):
            r = bad[0][0]
            print("  query", r, "gpu idx", gi[r][:12], "\n  oracle ", ei[r][:12])
            print("  gpu s64", top.scores_f64[r][:6].cpu().numpy(), "\n  oracle ", es[r][:6])
            full = (qs.double() @ db.double().T)[r].cpu().numpy()
            print("  full[gpu idx]", full[gi[r][:6]], " full[oracle idx]", full[ei[r][:6]])
            print("  element 13*j%64 beyond d=30 touched:", [(13 * j) % 64 for j in range(8)])
            break
