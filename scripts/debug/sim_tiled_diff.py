"""Where do the two routes of dlc_sdav_similarity_matrix differ on the tiled real frames (fan_in weights)?"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deeploopcloser_amd as dlc
import config1_common as c1
from oracle import similarity as osim
eng = dlc.default_engine()
paths = c1.frame_paths()
parser = dlc.CvInputParser(30, 41)
x = parser.parse_batch(np.stack([dlc.read_ppm(p) for p in paths]))
rng = np.random.RandomState(8)
tiles = [x]
for c in range(10):
    t = x.clone()
    if c >= 2:
        for f in range(20):
            for _ in range(1 + c):
                t[f, rng.randint(30), rng.randint(1681)] += (1.0 if rng.rand() < 0.5 else -1.0) / 255.0
        t.clamp_(0.0, 1.0)
    if c >= 6:
        t[:, 3] = 0.0; t[:, 4] = 0.0; t[::2, 9] = t[::2, 8]
    tiles.append(t)
xs = torch.cat(tiles)
n = xs.shape[0]
net = dlc.SDAV(seed=c1.SEED, weight_scale="fan_in")
ds = net.transform_tensor(xs).reshape(n, 30, 2500)
score = eng.distinctive_score(ds, 0.5, 0.2)
f_i8, _ = (t.clone() if t is not None else None for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False))
f_64, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False, force_f64=True)
a, b = f_i8.cpu().numpy(), f_64.cpu().numpy()
diff = np.argwhere((a != b) & ~(np.isnan(a) & np.isnan(b)))
print("differing entries:", len(diff))
dsn, sc = ds.cpu().numpy(), score.cpu().numpy()
seen = 0
for i, j in diff:
    if i >= j: continue
    idx = osim.match_features(dsn[i], dsn[j])
    d = osim.weighted_distances(dsn[i], dsn[j], idx, sc)
    with np.errstate(divide="ignore"):
        want = np.sum(10 - 10 * np.log(d))
    print(i, j, "i8", repr(a[i, j]), "f64", repr(b[i, j]), "oracle", repr(want), "min d", d.min())
    seen += 1
    if seen >= 12: break
print("value range of ds:", float(ds.min()), float(ds.max()))
