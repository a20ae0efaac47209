"""Randomised stress of the cosine top-k path (GPU box only; not part of the test suite): random database sizes,
descriptor widths, query counts, k and storage types across every plan (bandwidth kernel for <= 4 queries, MFMA tile,
split-K, small-database plan, multi-workgroup re-score), each checked against an fp64 product of the STORED rows on the GPU
-- EXACTLY, as the contract says (include/dlc.h):
  * returned rows == the top-k of the fp64 scores ordered by round(s * 2^40) descending, ties -> lower index;
  * returned fp64 scores == those scores to 1e-12, the fp32 ones their rounding;
  * status in {0, 2}; row shards + fp64 merge == the unsharded call bit for bit (indices and scores), whatever the plans.
Crowded cases (many rows one ulp apart) are mixed in so that the exhaustive pass runs.
Usage: python scripts/stress_cosine.py [seconds, default 60] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = dlc.default_engine()
rng = np.random.RandomState(seed)
g = torch.Generator(device=eng.device); g.manual_seed(seed)
t0, cnt, exhaustive, crowded, limited = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.randint(1, 300), rng.randint(300, 20000), rng.randint(20000, 120000)]))
    d = int(rng.choice([rng.randint(1, 200), 64 * rng.randint(1, 20), rng.randint(200, 3000)]))
    q = int(rng.choice([1, 2, 3, 4, 5, rng.randint(6, 300), rng.randint(300, 700)]))
    k = int(min(rng.choice([1, 5, 20, rng.randint(1, 129)]), 128))
    dtype = "bf16" if rng.rand() < 0.6 else "f16"
    if n * d > 1.5e8:
        continue
    x = torch.randn((n, d), generator=g, device=eng.device)
    if n > 10 and rng.rand() < 0.3:
        x[torch.from_numpy(rng.randint(1, n, size=min(n, 7))).to(eng.device)] = x[0].clone()   # exact duplicates: ties
    db = eng.normalize(x, dtype)
    if n > 400 and rng.rand() < 0.25:                                   # near-duplicates one ulp apart, spread over many groups
        crowded += 1
        bits = db.view(torch.int16)
        where = rng.choice(n, size=min(n // 4, 200), replace=False)
        for j, r in enumerate(where.tolist()):
            bits[r] = bits[0]
            c = (13 * j) % d                                            # (a real element: +-1 on a padding zero would make a NaN)
            if int(bits[r, c]) & 0x7fff not in (0, 0x7f80 - 1, 0x7c00 - 1):
                bits[r, c] += 1 if j % 2 else -1
    qs = eng.normalize(torch.randn((q, d), generator=g, device=eng.device) + (x[:q] if q <= n and rng.rand() < 0.5 else 0), dtype)
    if q <= n and rng.rand() < 0.5:
        qs[0] = db[0]
    top = eng.match_topk(qs, db, k, details=True)
    kk = min(k, n)
    full = qs.double() @ db.double().T                                  # [q, n] fp64 scores of the stored values
    key = torch.round(full * 2.0 ** 40)
    order = torch.sort(-key, dim=1, stable=True).indices[:, :kk]        # key descending, ties -> lower index
    assert torch.equal(top.idx[:, :kk], order), ("indices", n, d, q, k, dtype, int((top.idx[:, :kk] != order).sum()))
    assert bool((top.idx[:, kk:] == -1).all())
    want = torch.gather(full, 1, order)
    assert float((top.scores_f64[:, :kk] - want).abs().max()) < 1e-12, ("f64 scores", n, d, q, k)
    assert torch.equal(top.scores[:, :kk], top.scores_f64[:, :kk].float())
    st = set(top.status.cpu().tolist())
    assert st <= {0, 2}, st
    exhaustive += int((top.status == 2).sum())
    if n >= 16 and rng.rand() < 0.5:
        parts = int(rng.choice([2, 3, 8]))
        ps, pi = [], []
        for r in range(parts):
            lo, hi = dlc.shard_bounds(n, parts, r)
            t = eng.match_topk(qs, db[lo:hi], k, row_offset=lo, details=True)
            ps.append(t.scores_f64.clone()); pi.append(t.idx.clone())
        m = eng.topk_merge(torch.stack(ps), torch.stack(pi), details=True)
        assert torch.equal(m.idx, top.idx) and torch.equal(m.scores_f64, top.scores_f64) and torch.equal(m.scores, top.scores), \
            ("sharded", n, d, q, k, parts)
    if rng.rand() < 0.4:
        # the age-limited match (dlc_cosine_topk_older): query i sees the rows below limit0 + i only
        limit0 = int(rng.choice([rng.randint(-q, n + 1), n - q, rng.randint(0, 20), n + 5]))
        old = eng.match_topk(qs, db, k, details=True, older_than=limit0)
        lim = (limit0 + torch.arange(q, device=eng.device)).clamp(0, n)
        hidden = torch.arange(n, device=eng.device).unsqueeze(0) >= lim.unsqueeze(1)
        mkey = torch.where(hidden, torch.full_like(key, float("-inf")), key)
        morder = torch.sort(-mkey, dim=1, stable=True).indices[:, :kk]
        seen = torch.gather(~hidden, 1, morder)
        want_i = torch.where(seen, morder, torch.full_like(morder, -1))
        assert torch.equal(old.idx[:, :kk], want_i), ("older: indices", n, d, q, k, dtype, limit0)
        assert bool((old.idx[:, kk:] == -1).all())
        want_s = torch.where(seen, torch.gather(full, 1, morder), torch.full_like(want, float("-inf")))
        got_s = old.scores_f64[:, :kk]
        assert torch.equal(got_s.isinf(), want_s.isinf()) and float(torch.nan_to_num(got_s - want_s, nan=0.0).abs().max()) < 1e-12, \
            ("older: scores", n, d, q, k, limit0)
        assert set(old.status.cpu().tolist()) <= {0, 2}
        limited += 1
    cnt += 1
torch.cuda.synchronize()
print("cosine top-k: %d random cases exact (%d crowded; %d queries resolved by the exhaustive pass; %d cases also with an age limit per query)"
      % (cnt, crowded, exhaustive, limited), flush=True)
