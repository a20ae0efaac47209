"""GPU parity of the streaming loop-closure queries and of the growable
key-frame database they run on, against oracle/loop_closure.py on the rows as stored.
Indices are compared exactly except where the oracle's own scores are closer than fp32 can
resolve (2e-6); scores to 2e-5 as for every cosine test."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dlc():
    import deeploopcloser_amd as d
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    d.default_engine()
    return d


def revisiting_sequence(t, d, period, seed):
    """A trajectory that returns to each place every `period` frames (plus noise)."""
    rng = np.random.RandomState(seed)
    places = rng.standard_normal((period, d)).astype(np.float32)
    return places[np.arange(t) % period] + 0.3 * rng.standard_normal((t, d)).astype(np.float32)


def check_against_oracle(det, s, i, k, exclusion):
    from oracle import loop_closure as oloop
    rows = det.db.rows.float().cpu().numpy().astype(np.float64)
    es, ei = oloop.stream_topk(rows, k, exclusion)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    assert s.shape == es.shape and i.dtype == np.int64
    assert np.array_equal(i == -1, ei == -1) and np.array_equal(np.isneginf(s), np.isneginf(es))
    ok = ei >= 0
    assert np.abs(s[ok] - es[ok]).max() < 2e-5                   # (the scores come back as fp32)
    assert np.array_equal(i, ei)                                  # the order is decided on the fp64 scores: index for index


@pytest.mark.parametrize("batch,exclusion,k,dtype", [(1, 0, 3, "bf16"), (7, 3, 5, "bf16"), (50, 40, 5, "f16"),
                                                     (200, 10, 20, "bf16"), (64, 0, 128, "bf16")])
def test_stream_equals_oracle_for_any_batching(dlc, batch, exclusion, k, dtype):
    t, d = 330, 192
    x = revisiting_sequence(t, d, 97, seed=batch)
    det = dlc.LoopClosureDetector(d, k=k, threshold=0.5, exclusion=exclusion, dtype=dtype, capacity=16)
    out_s, out_i = [], []
    for lo in range(0, t, batch):
        s, i = det.query_and_insert(x[lo:lo + batch])
        out_s.append(s)
        out_i.append(i)
    s, i = torch.cat(out_s), torch.cat(out_i)
    assert len(det) == t and det.db.capacity >= t              # grew from 16 by doubling
    check_against_oracle(det, s, i, k, exclusion)
    # every frame old enough to have seen its place before finds that earlier visit first
    first = i[:, 0].cpu().numpy()
    for g in range(97 + exclusion + 1, t):
        assert first[g] % 97 == g % 97
    loops = det.loops(s, i, 0)
    assert loops and all(sc >= 0.5 and m < g - exclusion for g, m, sc in loops)


def test_empty_batch_and_errors(dlc):
    det = dlc.LoopClosureDetector(64, k=4, exclusion=2)
    s, i = det.query_and_insert(np.zeros((0, 64), dtype=np.float32))
    assert s.shape == (0, 4) and i.shape == (0, 4) and len(det) == 0
    with pytest.raises(ValueError):
        det.query_and_insert(np.zeros((3, 200), dtype=np.float32))      # wrong width
    with pytest.raises(ValueError):
        dlc.LoopClosureDetector(64, k=0)
    with pytest.raises(ValueError):
        dlc.LoopClosureDetector(64, k=129)
    with pytest.raises(ValueError):
        dlc.LoopClosureDetector(64, exclusion=-1)


def test_appended_database_equals_one_built_at_once(dlc):
    rng = np.random.RandomState(5)
    x = rng.standard_normal((1000, 100)).astype(np.float32)
    q = rng.standard_normal((9, 100)).astype(np.float32)
    whole = dlc.KeyframeDatabase(x, dtype="bf16", center=True, row_offset=50)
    grown = dlc.KeyframeDatabase.empty(100, capacity=3, dtype="bf16", center=True, row_offset=50)
    ids = [grown.append(x[lo:lo + 333]) for lo in range(0, 1000, 333)]
    assert ids == [(50, 383), (383, 716), (716, 1049), (1049, 1050)]
    assert len(grown) == 1000 and torch.equal(grown.rows, whole.rows)
    s0, i0 = whole.match_topk(q, 10)
    s1, i1 = grown.match_topk(q, 10)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    part = grown.prefix(400)
    assert len(part) == 400 and part.rows.data_ptr() == grown.rows.data_ptr()
    s2, i2 = part.match_topk(q, 10)
    s3, i3 = dlc.KeyframeDatabase(x[:400], dtype="bf16", center=True, row_offset=50).match_topk(q, 10)
    assert torch.equal(i2, i3) and torch.equal(s2, s3)
    with pytest.raises(ValueError):
        grown.prefix(1001)
    with pytest.raises(ValueError):
        grown.append(np.zeros((2, 300), dtype=np.float32))


def test_cli_streams_the_reference_frames(dlc, capsys):
    from deeploopcloser_amd import loop_closure
    rc = loop_closure.main([os.path.join(GOLDEN, "frames"), "--network", "cnn_vtl", "--k", "2", "--exclusion", "0",
                            "--threshold", "-1", "--batch", "2"])
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0
    # frame 1 sees frame 0; frame 2 sees frames 0 and 1 -> three candidate lines
    got = sorted((int(l.split("\t")[1]), int(l.split("\t")[3])) for l in out)
    assert got == [(1, 0), (2, 0), (2, 1)] and all(l.startswith("loop\t") for l in out)
    rc = loop_closure.main([os.path.join(GOLDEN, "frames"), "--network", "sdav", "--k", "1", "--exclusion", "0",
                            "--threshold", "-1", "--batch", "3"])
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 2


def test_describe_sdav_takes_a_chunk_of_frames_of_several_sizes(dlc, tmp_path):
    """The reference parses frames one by one (CvInputParser.py:30-33), so a dataset may mix sizes: a chunk of two sizes
    gives, frame for frame, the descriptors each frame gets in a chunk of its own."""
    import glob
    import numpy as np
    from deeploopcloser_amd import loop_closure
    from deeploopcloser_amd.input import read_ppm
    src = sorted(glob.glob(os.path.join(GOLDEN, "frames", "*.ppm")))[:2]
    a, b = read_ppm(src[0]), read_ppm(src[1])[:160, :200]
    files = []
    for name, img in (("a.ppm", a), ("b.ppm", np.ascontiguousarray(b))):
        f = str(tmp_path / name)
        with open(f, "wb") as fh:
            fh.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
        files.append(f)
    net = dlc.SDAV()
    both = loop_closure.describe_sdav(files, net)
    one_a = loop_closure.describe_sdav(files[:1], net)
    one_b = loop_closure.describe_sdav(files[1:], net)
    assert tuple(both.shape) == (2, 30 * 2500)
    assert torch.equal(both[0], one_a[0]) and torch.equal(both[1], one_b[0])


def test_keep_older_kernel_equals_torch_form(dlc):
    """dlc_topk_keep_older == first_k_eligible (its torch form, tested on the CPU) on random candidate lists: empty
    slots, every candidate too recent, fewer candidates than k, more than 64 candidates per row."""
    from deeploopcloser_amd.loop_closure import first_k_eligible
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device); g.manual_seed(5)
    for (b, kk, k, limit0) in ((1, 1, 1, 0), (7, 11, 5, 3), (32, 36, 5, 100), (200, 128, 20, -50), (64, 100, 128, 40), (3, 5, 8, 2)):
        idx = torch.randint(-1, 160, (b, kk), generator=g, device=eng.device, dtype=torch.int64)
        sc = torch.rand((b, kk), generator=g, device=eng.device, dtype=torch.float32).sort(dim=1, descending=True).values
        limit = torch.arange(limit0, limit0 + b, device=eng.device)
        ws, wi = first_k_eligible(sc, idx, limit, k)
        gs, gi = eng.topk_keep_older(sc, idx, limit0, k)
        assert torch.equal(gi, wi) and torch.equal(gs, ws), (b, kk, k, limit0)


def test_match_with_age_limit_equals_wide_match_then_keep_older(dlc):
    """dlc_cosine_topk_older (query i sees rows below limit0 + i) == dlc_cosine_topk with k + q - 1 candidates followed by
    dlc_topk_keep_older, index for index and score bit for score bit, on every plan of the match: the small-database plan,
    the one-pass plan, few queries with long rows (re-score spread over workgroups + merge); limits that leave some queries
    nothing, limits past the database, limits inside a group of 8 rows; databases full of exact copies (ties)."""
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device); g.manual_seed(11)
    cases = [(32, 700, 192, 5, 650, 0), (32, 700, 192, 5, -10, 0), (7, 100, 64, 3, 97, 0), (5, 300, 33024, 4, 283, 0),
             (64, 70000, 128, 8, 69950, 0), (32, 500, 64, 6, 470, 50), (40, 333, 256, 20, 1000, 0), (3, 9, 64, 2, 1, 3),
             (16, 700, 192, 5, -40, 0), (32, 70000, 128, 5, 300, 0)]
    for (q, n, d, k, limit0, distinct) in cases:
        x = torch.randn((n, d), generator=g, device=eng.device, dtype=torch.float32)
        if distinct:
            x = x[torch.randint(0, distinct, (n,), generator=g, device=eng.device)]
        db = eng.normalize(x, torch.bfloat16, False)
        qs = eng.normalize(x[torch.randint(0, n, (q,), generator=g, device=eng.device)] +
                           0.2 * torch.randn((q, d), generator=g, device=eng.device, dtype=torch.float32), torch.bfloat16, False)
        # the two-step form: the rows the NEWEST query may see, k + q - 1 candidates (at most q - 1 of them too recent for a
        # query), then every query's first k that are old enough
        n_newest = max(0, min(n, limit0 + q - 1))
        if n_newest > 0:
            ws, wi = eng.match_topk(qs, db[:n_newest], min(k + q - 1, 128))
            ws, wi = eng.topk_keep_older(ws, wi, limit0, k)
        else:
            ws = torch.full((q, k), float("-inf"), dtype=torch.float32, device=eng.device)
            wi = torch.full((q, k), -1, dtype=torch.int64, device=eng.device)
        got = eng.match_topk(qs, db, k, older_than=limit0, details=True)
        assert torch.equal(got.idx, wi), (q, n, d, k, limit0)
        assert torch.equal(got.scores, ws), (q, n, d, k, limit0)
        assert int(got.status.max()) in (0, 2)
        with_off = eng.match_topk(qs, db, k, row_offset=1000, older_than=limit0)
        assert torch.equal(with_off[1], torch.where(wi >= 0, wi + 1000, wi)), (q, n, d, k, limit0)


# ---------------------------------------------------------------------------------- reference-semantics streaming query
def test_similarity_stream_rows_equal_matrix_columns():
    """SimilarityStream (dlc_sdav_stream_*): for every frame f that arrives, row[j] = score(h_j, h_f) for all older j ==
    column f of the all-vs-all matrix call above the diagonal, BIT FOR BIT -- arg-min decided by the same filter (a resident
    panel quantised over the fixed range (0, 1), never re-quantised), same terms, same summation order.  Saturated data
    with copies of patches and of frames (+inf scores, exact ties), near-ties one ulp apart, several shapes; frames appended
    singly and in batches, with the stream growing past its capacity on the way."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device)
    g.manual_seed(4)
    for n, p, h in ((60, 30, 250), (45, 30, 2500), (70, 7, 129), (40, 32, 64), (33, 13, 1000)):
        ds = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
        ds[5] = ds[2]; ds[n - 1] = ds[n - 3]                       # copies of whole frames
        ds[7, 1] = ds[7, 0]                                        # a patch twice inside a frame
        if p > 4:
            ds[9, 3] = ds[9, 2]; ds[9, 3, 0] += 1e-9                # two patches one step apart: a near-tie for everybody
        ds.clamp_(0.0, 1.0)
        score = eng.distinctive_score(ds, 0.5, 0.2)
        mf, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)
        st = dlc.SimilarityStream(score, patches=p, width=h, capacity=16)
        f = 0
        for b in (1, 1, 3, 1, 8, 1, n):                             # arrival pattern: singles and batches
            take = min(b, n - f)
            if take <= 0:
                break
            st.append(ds[f:f + take])
            for q in range(f, f + take):
                row = st.query(q)
                assert row.shape == (q,)
                assert torch.equal(torch.nan_to_num(row, posinf=1e300), torch.nan_to_num(mf[:q, q], posinf=1e300)), (n, p, h, q)
                assert torch.equal(row.isinf(), mf[:q, q].isinf())
            f += take
        assert len(st) == n and st.capacity >= n and int(st.stats[1]) == 0
        one = st.query()                                           # default: the newest frame
        assert torch.equal(torch.nan_to_num(one, posinf=1e300), torch.nan_to_num(mf[:n - 1, n - 1], posinf=1e300))


def test_similarity_stream_column_centres_low_contrast():
    """A stream whose range is given per column (col_centre + a narrow value_range): low-contrast descriptors -- columns
    within ~1e-3 of their own means, the means spread over [0.15, 0.88] -- arrive one by one; every row == the matrix call's
    column bit for bit, and the int8 products decide nearly every arg-min (over the plain range (0, 1) the error window
    would swallow them all: same rows, all of them through the direct evaluation)."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device)
    g.manual_seed(9)
    n, p, h = 48, 30, 2500
    means = 0.15 + 0.73 * torch.rand((h,), generator=g, device=eng.device, dtype=torch.float64)
    ds = means + 1e-3 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
    score = eng.distinctive_score(ds, 0.5, 0.2)
    mf, _ = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)
    dev = float((ds - means).abs().max())
    counts = {}
    for name, kw in (("centred", dict(value_range=(-1.05 * dev, 1.05 * dev), column_centre=means)), ("plain", dict())):
        st = dlc.SimilarityStream(score, patches=p, width=h, capacity=64, **kw)
        st.append(ds)
        direct = 0
        for q in range(1, n):
            row = st.query(q)
            assert torch.equal(row, mf[:q, q]), (name, q)
            assert int(st.stats[1]) == 0
            direct += int(st.stats[0])
        counts[name] = direct
    total = n * (n - 1) // 2 * p
    print("stream, low-contrast columns: %d of %d arg-mins evaluated directly with column centres, %d over (0, 1)"
          % (counts["centred"], total, counts["plain"]))
    assert counts["centred"] <= 0.02 * total and counts["plain"] >= 0.9 * total


def test_similarity_stream_vs_oracle_on_real_frames_and_poison():
    """The 20 real frames of datasets/test through the GPU front-end and SDAV.transform, fed to the stream one by one:
    every row against the oracle's similarity_score (the reference's arithmetic restated, oracle/similarity.py); the
    detector's top-k; and a frame with a value outside the stream's range poisons it LOUDLY (NaN + stats[1])."""
    import deeploopcloser_amd as dlc
    import config1_common as c1
    from oracle import similarity as osim
    eng = dlc.default_engine()
    paths = c1.frame_paths()
    x = dlc.CvInputParser(30, 41).parse_batch(np.stack([dlc.read_ppm(p_) for p_ in paths]))
    h = dlc.SDAV(seed=c1.SEED).transform_tensor(x).reshape(len(paths), 30, 2500)
    hn = h.cpu().numpy()
    sc = osim.distinctive_score(osim.average_response(hn))
    det = dlc.SdavLoopClosureDetector(h, k=3, exclusion=2, capacity=4)
    np.testing.assert_allclose(det.stream.score.cpu().numpy(), sc, rtol=1e-12)
    for f in range(len(paths)):
        s, i = det.query_and_insert(h[f])
        want = np.array([np.sum(10 - 10 * np.log(osim.weighted_distances(hn[j], hn[f], osim.match_features(hn[j], hn[f]), sc)))
                         for j in range(f)])
        row = det.stream.query(f).cpu().numpy()
        assert row.shape == want.shape and (f == 0 or np.abs(row - want).max() <= 1e-9 * np.abs(want).max())
        n_see = f - 2
        if n_see > 0:
            order = np.lexsort((np.arange(n_see), -want[:n_see]))[:3]
            assert i[0, :len(order)].cpu().tolist() == order.tolist()
            assert np.allclose(s[0, :len(order)].cpu().numpy(), want[order], rtol=1e-9)
        else:
            assert int(i.max()) == -1
    bad = h[3].clone()
    bad[4, 7] = 1.25                                              # outside (0, 1): the fixed-point bound does not cover it
    row = det.stream.query_and_insert(bad)
    assert bool(row.isnan().all()) and int(det.stream.stats[1]) == 1
    s, i = det.query_and_insert(h[5])                             # ... and the detector's lists say so themselves: (NaN, -1)
    assert int(i.max()) == -1 and bool(s.isnan().all()) and int(det.poisoned) == 1
    with pytest.raises(RuntimeError, match="outside the stream's fixed range"):
        det.loops(s, i, len(det) - 1)


def test_topk_rows_f64_and_batched_detector():
    """dlc_topk_rows_f64 (the detector's ranking: score descending, ties -> the older frame, NaN never, per-row limits)
    against a stable sort; and SdavLoopClosureDetector.query_and_insert of a BATCH == frame by frame."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device)
    g.manual_seed(12)
    rows, ld, k = 37, 1500, 7
    sc = torch.randn((rows, ld), generator=g, device=eng.device, dtype=torch.float64)
    sc[:, ::5] = sc[:, 1::5][:, :sc[:, ::5].shape[1]]              # exact ties
    sc[3, :] = 2.5                                                  # a whole row of ties
    sc[4, 10:900] = float("nan"); sc[5, :] = float("nan"); sc[6, 17] = float("inf"); sc[7, 3] = float("-inf")
    clean = torch.zeros(1, dtype=torch.int64, device=eng.device)
    s, i = eng.topk_rows_f64(sc, ld, 0, k, poison=clean + 3)       # the poison word: (NaN, -1) everywhere
    assert bool(s.isnan().all()) and bool((i == -1).all())
    for limit0, step in ((ld, 0), (-3, 1), (4, 40), (0, 0)):
        s, i = eng.topk_rows_f64(sc, limit0, step, k, poison=clean if step else None)
        for r in range(rows):
            n = max(0, min(ld, limit0 + r * step))
            v = sc[r, :n].cpu().numpy()
            ok = np.nonzero(~np.isnan(v))[0]
            order = ok[np.lexsort((ok, -v[ok]))][:k]
            want_i = np.full(k, -1, dtype=np.int64); want_i[:len(order)] = order
            want_s = np.full(k, -np.inf); want_s[:len(order)] = v[order]
            assert np.array_equal(i[r].cpu().numpy(), want_i) and np.array_equal(s[r].cpu().numpy(), want_s), (limit0, step, r)
    n, p, h = 40, 30, 250
    ds = torch.sigmoid(4.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    ds[9] = ds[2]
    one = dlc.SdavLoopClosureDetector(ds, patches=p, width=h, k=4, exclusion=3, capacity=8)
    bat = dlc.SdavLoopClosureDetector(ds, patches=p, width=h, k=4, exclusion=3, capacity=8)
    parts = [one.query_and_insert(ds[f]) for f in range(n)]
    s1, i1 = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
    outs, f = [], 0
    for b in (1, 6, 1, 17, n):
        take = min(b, n - f)
        if take > 0:
            outs.append(bat.query_and_insert(ds[f:f + take]))
            f += take
    s2, i2 = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    assert int(i1[9, 0]) == 2 and bool(torch.isinf(s1[9, 0]))       # the repeated frame finds its first sighting
    # dlc_sdav_stream_query_batch: rows of several resident frames in one pair of launches == one query each, bit for bit
    st = bat.stream
    for first, count in ((0, 1), (0, 5), (1, 1), (7, 13), (n - 3, 3), (0, n)):
        rows = st.query_batch(first, count)
        assert rows.shape == (count, max(1, first + count - 1))
        for q in range(count):
            f = first + q
            want = st.query(f)
            assert torch.equal(torch.nan_to_num(rows[q, :f], posinf=1e300), torch.nan_to_num(want, posinf=1e300)), (first, count, q)
            assert torch.equal(rows[q, :f].isinf(), want.isinf())
    with pytest.raises(ValueError):
        st.query_batch(n - 1, 2)
    assert one.loops(s1, i1, 0) == bat.loops(s2, i2, 0)


def test_batched_stream_query_as_a_strip_of_the_matrix_call():
    """Batches of 8 frames and more go through the all-vs-all call's product kernel (dlc_sdav_stream_query_batch: the
    batch's frames are a strip of its columns): rows == single queries == the matrix call's columns, bit for bit -- for
    batches that start anywhere in a block of 32 frames and span one, two or three of them, with copies of patches and of
    frames (exact ties, +inf scores), near-copies one ulp apart, and a stream that has grown past its first capacity."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device)
    g.manual_seed(21)
    n, p, h = 150, 30, 250
    ds = torch.sigmoid(6.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    ds[70] = ds[11]                                                  # a frame seen twice
    ds[71, 4] = ds[71, 3]                                            # a patch twice in one frame
    ds[90, 7] = torch.nextafter(ds[90, 6], torch.ones_like(ds[90, 6]))   # one ulp apart
    ds[120:124, 5] = ds[40, 5]                                       # one patch in several frames
    score = eng.distinctive_score(ds, 0.5, 0.2)
    want = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, want_int64=False)[0]
    st = dlc.SimilarityStream(score, patches=p, width=h, capacity=16)
    st.append(ds[:100])
    st.append(ds[100:])
    stats_total = 0
    for first, count in ((0, 8), (0, 32), (5, 8), (31, 9), (37, 64), (96, 32), (64, 86), (142, 8), (1, 149)):
        rows = st.query_batch(first, count)
        stats_total += int(st.stats[0])
        for q in range(count):
            f = first + q
            got, col = rows[q, :f], want[:f, f]
            assert torch.equal(got.isinf(), col.isinf()), (first, count, q)
            assert torch.equal(torch.nan_to_num(got, posinf=1e300), torch.nan_to_num(col, posinf=1e300)), (first, count, q)
    assert bool(torch.isinf(want[11, 70])) and stats_total > 0       # the copies went through the direct evaluation
    one = st.query(77)
    assert torch.equal(torch.nan_to_num(one, posinf=1e300), torch.nan_to_num(want[:77, 77], posinf=1e300))
    # other patch counts: 7 patches = nine frames per 64-column unit, 16 = four, 32 = two (no padding columns)
    for p2, h2, n2 in ((7, 96, 90), (16, 130, 70), (32, 64, 45)):
        d2 = torch.sigmoid(5.0 * torch.randn((n2, p2, h2), generator=g, device=eng.device, dtype=torch.float64))
        d2[n2 // 2] = d2[3]
        sc2 = eng.distinctive_score(d2, 0.5, 0.2)
        w2 = eng.sdav_similarity_matrix(d2, sc2, 10.0, -10.0, want_int64=False)[0]
        s2 = dlc.SimilarityStream(sc2, patches=p2, width=h2, capacity=n2)
        s2.append(d2)
        for first, count in ((0, n2), (11, min(40, n2 - 11)), (n2 - 9, 9)):
            rows = s2.query_batch(first, count)
            for q in range(count):
                f = first + q
                got, col = rows[q, :f], w2[:f, f]
                assert torch.equal(got.isinf(), col.isinf()), (p2, first, count, q)
                assert torch.equal(torch.nan_to_num(got, posinf=1e300), torch.nan_to_num(col, posinf=1e300)), (p2, first, count, q)


def test_detector_two_batches_in_flight_equals_batch_by_batch():
    """SdavLoopClosureDetector.submit / result: a batch's copy + quantisation beside the previous batch's product kernel, its
    resolution + scores + ranking beside the next batch's (two streams, dlc_sdav_stream_query_batch_staged) -- the lists are
    query_and_insert's bit for bit, for batch sizes on both sides of the strip's threshold, a stream that grows under the
    pipeline, copies of frames (ties, +inf scores), results fetched late (one ticket behind) and at once."""
    import deeploopcloser_amd as dlc
    eng = dlc.default_engine()
    g = torch.Generator(device=eng.device)
    g.manual_seed(31)
    n, p, h = 150, 30, 300
    ds = torch.sigmoid(4.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
    ds[40] = ds[7]
    ds[41, 3] = ds[8, 11]
    ds[120] = ds[119]
    for sizes, late in (((32, 32, 32, 32, 22), True), ((9, 3, 40, 8, 1, 16, 33, 40), True), ((16, 16, 64, 54), False)):
        plain = dlc.SdavLoopClosureDetector(ds, patches=p, width=h, k=4, exclusion=2, capacity=48)
        piped = dlc.SdavLoopClosureDetector(ds, patches=p, width=h, k=4, exclusion=2, capacity=48)
        want, got, f, prev = [], [], 0, None
        for b in sizes:
            chunk = ds[f:f + b]
            want.append(plain.query_and_insert(chunk))
            t = piped.submit(chunk)
            if late:
                if prev is not None:
                    got.append(piped.result(prev))
                prev = t
            else:
                got.append(piped.result(t))
            f += b
        if late:
            got.append(piped.result(prev))
        assert f == n and len(piped) == n
        ws, wi = torch.cat([w[0] for w in want]), torch.cat([w[1] for w in want])
        gs, gi = torch.cat([o[0] for o in got]), torch.cat([o[1] for o in got])
        torch.cuda.synchronize()
        assert torch.equal(wi, gi), sizes
        assert torch.equal(torch.nan_to_num(ws, posinf=1e300, neginf=-1e300), torch.nan_to_num(gs, posinf=1e300, neginf=-1e300)), sizes
        assert int(gi[40, 0]) == 7 and bool(torch.isinf(gs[40, 0]))
        assert int(piped.stream.stats[1]) == 0
    with pytest.raises(ValueError):
        piped.result(0)                                               # long gone
    before = piped._tickets
    with pytest.raises(ValueError):
        piped.submit(ds[:9, :, :7])                                   # a wrong width: refused before a ticket is spent
    assert piped._tickets == before and len(piped) == n
    # the two halves of the strip form through the engine: stage 1 + stage 2 == the one call
    st = piped.stream
    first, b = 100, 32
    need = eng.lib.dlc_sdav_stream_query_batch_workspace_bytes(st.capacity, st.p, b)
    ws_ = torch.empty(int(need), dtype=torch.uint8, device=eng.device)
    rows = torch.zeros((b, st.capacity), dtype=torch.float64, device=eng.device)
    eng.sdav_stream_query_batch_staged(st.state, st.desc, first, b, st.score, 1, rows, ws_, st.a, st.b)
    eng.sdav_stream_query_batch_staged(st.state, st.desc, first, b, st.score, 2, rows, ws_, st.a, st.b)
    one = st.query_batch(first, b)
    for q in range(b):
        assert torch.equal(torch.nan_to_num(rows[q, :first + q], posinf=1e300), torch.nan_to_num(one[q, :first + q], posinf=1e300))
    with pytest.raises(Exception):
        eng.sdav_stream_query_batch_staged(st.state, st.desc, first, 4, st.score, 1, rows[:4], ws_, st.a, st.b)   # no strip below 8
