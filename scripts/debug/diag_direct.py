"""Which arg-mins of the tiled-real-frames similarity are evaluated directly, and with how many candidates (GPU box):
emulates the filter's decision in NumPy on the frame pairs the call marks in direct_pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import deeploopcloser_amd as dlc
import real_frames
eng = dlc.default_engine()
n, p, h = int(sys.argv[1]) if len(sys.argv) > 1 else 220, 30, 2500
xs = real_frames.tiled_patches(dlc, n)
ds = dlc.SDAV(seed=4, weight_scale="fan_in").transform_tensor(xs).reshape(n, p, h)
score = eng.distinctive_score(ds, 0.5, 0.2)
stats = torch.zeros((2,), dtype=torch.int64, device=eng.device)
dmap = torch.zeros((n, n), dtype=torch.uint8, device=eng.device)
eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, stats=stats, direct_pairs=dmap)
print("stats", stats.tolist())
d = ds.cpu().numpy()
flat = d.reshape(-1, h)
cmin, cmax = flat.min(0), flat.max(0)
s = (cmax - cmin).max()
print("s", s, "median col range", np.median(cmax - cmin), "argmax col", int(np.argmax(cmax - cmin)))
rng_sorted = np.sort(cmax - cmin)
print("col ranges: top 10", rng_sorted[-10:], "p50", rng_sorted[h // 2])
v = (flat - (cmin + 0.5 * (cmax - cmin))) * (0.996 / s)
sv = np.abs(v).sum(1).max()
ed = 2.0**-23 * sv + h * (2.0**-24 + 2.0**-33 + 2.0**-46) + 1.004 * 2.0**-15 + 2.0**-16
win = 2 * ed + 1e-8
print("sv", sv, "window", win)
V = v.reshape(n, p, h)
pairs = np.argwhere(dmap.cpu().numpy() != 0)
print("pairs", len(pairs))
rs = np.random.RandomState(0)
sel = pairs[rs.permutation(len(pairs))[:400]]
hist_a = np.zeros(p, int); ncs = []; gaps = []; cells = []
for i, j in sel:
    d2 = ((V[i][:, None, :] - V[j][None, :, :]) ** 2).sum(-1)
    keep = np.ones(p, bool)
    for b in range(p):
        for e in range(b):
            if keep[e] and np.array_equal(V[j][e], V[j][b]): keep[b] = False; break
    d2k = d2[:, keep]
    srt = np.sort(d2k, axis=1)
    und = (srt[:, 1] - srt[:, 0]) <= win
    for a in np.nonzero(und)[0]:
        hist_a[a] += 1
        ncs.append(int(((d2k[a] - srt[a, 0]) <= win).sum()))
        gaps.append(srt[a, 1] - srt[a, 0])
        kb = np.nonzero(keep)[0]
        cand = kb[np.nonzero((d2k[a] - srt[a, 0]) <= win)[0]]
        cells.append((i // 20, i % 20, j // 20, j % 20, a, cand.tolist(), float(srt[a, 0]), float(srt[a, 1] - srt[a, 0]),
                      float(((V[j][cand[0]] - V[j][cand[1]]) ** 2).sum())))
print("undecided by a:", hist_a.tolist())
print("ncand: mean %.2f max %d hist %s" % (np.mean(ncs), max(ncs), np.bincount(ncs).tolist()))
print("gap median %.3g, best d2 median" % np.median(gaps))

import collections
print("by real frame j%20:", sorted(collections.Counter(c[3] for c in cells).items()))
print("by candidate pair (j%20, cands):", collections.Counter((c[3], tuple(c[5])) for c in cells).most_common(12))
for c in cells[:25]: print(c)
