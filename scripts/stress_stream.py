"""Randomised stress of the streaming similarity (GPU box only; not part of the test suite): random patch counts, widths,
frame counts and batch sizes; frames appended in random batches (the stream growing on the way), every batch queried with
dlc_sdav_stream_query_batch -- two frames per pass below 8 frames, a strip of the all-vs-all call's product kernel from 8
on -- and every row compared BIT FOR BIT with the matrix call's column (and a few with the single query).  Copies of
patches and of frames (+inf scores, exact ties) and near-copies one ulp apart are mixed in.
Usage: python scripts/stress_stream.py [cases, default 40] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = dlc.default_engine()
rng = np.random.RandomState(seed)
g = torch.Generator(device=eng.device); g.manual_seed(seed)
same = lambda a, b: torch.equal(a.isinf(), b.isinf()) and torch.equal(torch.nan_to_num(a, posinf=1e300), torch.nan_to_num(b, posinf=1e300))
rows_checked = strips = directs = 0
t0 = time.time()
for case in range(cases):
    p = int(rng.choice([1, 2, 7, 16, 30, 32, rng.randint(1, 33)]))
    h = int(rng.choice([8, 64, 250, 256, 300, rng.randint(2, 700)]))
    n = int(rng.choice([rng.randint(2, 40), rng.randint(40, 200), rng.randint(200, 420)]))
    if n * p * h > 4e6:
        n = max(2, int(4e6 / (p * h)))
    sharp = float(rng.choice([1.0, 6.0, 35.0]))
    ds = torch.sigmoid(sharp * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    for _ in range(rng.randint(0, 4)):                         # frames seen twice, patches twice, one ulp apart
        a, b = rng.randint(0, n, 2)
        ds[a] = ds[b]
    for _ in range(rng.randint(0, 4)):
        a, b = rng.randint(0, n, 2)
        ds[a, rng.randint(0, p)] = ds[b, rng.randint(0, p)]
    if rng.rand() < 0.5:
        a = rng.randint(0, n); pa, pb = rng.randint(0, p, 2)
        ds[a, pa] = torch.nextafter(ds[a, pb], torch.ones_like(ds[a, pb]))
    score = eng.distinctive_score(ds, 0.5, 0.2)
    want = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)[0]
    st = dlc.SimilarityStream(score, patches=p, width=h, capacity=int(rng.choice([1, 8, n])))
    f = 0
    while f < n:
        b = int(min(n - f, rng.choice([1, 2, 7, 8, 9, 31, 32, 33, 64, rng.randint(1, 100)])))
        first = st.append(ds[f:f + b])
        assert first == f
        rows = st.query_batch(f, b)
        directs += int(st.stats[0]); assert int(st.stats[1]) == 0
        strips += int(b >= 8)
        for q in range(b):
            assert same(rows[q, :f + q], want[:f + q, f + q]), ("batch", case, p, h, n, f, b, q)
        rows_checked += b
        if rng.rand() < 0.2 and f + b > 1:
            fq = f + rng.randint(0, b)
            assert same(st.query(fq), want[:fq, fq]), ("single", case, p, h, n, fq)
        f += b
    # a second pass over resident frames: batches that start anywhere
    for _ in range(3):
        first = rng.randint(0, n); cnt = int(min(n - first, rng.choice([1, 8, 20, 64])))
        rows = st.query_batch(first, cnt)
        for q in range(cnt):
            assert same(rows[q, :first + q], want[:first + q, first + q]), ("resident", case, p, h, n, first, cnt, q)
        rows_checked += cnt
    # r06: the detector with two batches in flight (submit / result: the strip's product kernel on a second stream, the small
    # kernels of the neighbouring batches beside it) against the detector batch by batch -- the same lists bit for bit
    k_, ex_ = int(rng.randint(1, 9)), int(rng.randint(0, 5))
    cap_ = int(rng.choice([1, 8, n]))
    plain = dlc.SdavLoopClosureDetector(score, patches=p, width=h, k=k_, exclusion=ex_, capacity=cap_)
    piped = dlc.SdavLoopClosureDetector(score, patches=p, width=h, k=k_, exclusion=ex_, capacity=cap_)
    f, prev, wl, gl = 0, None, [], []
    late = rng.rand() < 0.7
    while f < n:
        b = int(min(n - f, rng.choice([1, 7, 8, 9, 16, 31, 32, 33, 64, rng.randint(1, 100)])))
        wl.append(plain.query_and_insert(ds[f:f + b]))
        t_ = piped.submit(ds[f:f + b])
        if late:
            if prev is not None:
                gl.append(piped.result(prev))
            prev = t_
        else:
            gl.append(piped.result(t_))
        f += b
    if late:
        gl.append(piped.result(prev))
    ws_, wi_ = torch.cat([w_[0] for w_ in wl]), torch.cat([w_[1] for w_ in wl])
    gs_, gi_ = torch.cat([o_[0] for o_ in gl]), torch.cat([o_[1] for o_ in gl])
    torch.cuda.synchronize()
    assert torch.equal(wi_, gi_) and torch.equal(torch.nan_to_num(ws_, posinf=1e300, neginf=-1e300), torch.nan_to_num(gs_, posinf=1e300, neginf=-1e300)), \
        ("pipelined detector", case, p, h, n, k_, ex_, cap_, late)
    piped_rows = globals().get("piped_rows", 0) + n
torch.cuda.synchronize()
print("pipelined detector: %d frames' lists equal to the batch-by-batch detector's" % piped_rows, flush=True)
print("streaming similarity: %d cases, %d rows bit-identical to the matrix call's columns (%d batches as strips; %d direct evaluations) in %.0f s"
      % (cases, rows_checked, strips, directs, time.time() - t0), flush=True)
