"""Status of the age-limited match over a stream of batches (how often the selection's certificate fails)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
N, D, k, excl, b = 1063, 4096, 5, 30, 32
xs = torch.randn((N, D), generator=g, device=eng.device, dtype=torch.float32)
db = eng.normalize(xs, torch.bfloat16, False)
for lo in range(0, N, b):
    q = db[lo:lo + b]
    n_search = lo + q.shape[0] - 1 - excl
    if n_search <= 0:
        continue
    r = eng.match_topk(q, db[:n_search], k, older_than=lo - excl, details=True)
    st = r.status.cpu().numpy()
    if st.max() > 0:
        bad = [i for i in range(len(st)) if st[i]]
        print("batch at %d: n_search %d, status!=0 for queries %s" % (lo, n_search, bad), r.scores_f64[bad[0]].tolist(), r.idx[bad[0]].tolist(), flush=True)
print("done")
