"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs and against the committed golden fixtures.

Tolerances (stated here, used below):
  * integer / byte / index work (cnn_vtl distance, int8 descriptors, top-k
    indices on planted-margin data, argmin-driven similarity): bit-exact;
  * fp64 kernels (SDAV / DA / conv GEMMs, SDAV similarity): the reference is
    fp64 and so are the kernels; only the summation order differs -> abs 1e-10
    on sigmoid outputs, rel 1e-9 on similarity scores;
  * cosine top-k: the order is decided on fp64 scores of the stored values (as the oracle's) -> IDENTICAL
    indices, in every test including the fuzz loops; reported fp64 scores vs the oracle's BLAS fp64 -> abs
    1e-12; their fp32 roundings -> equal to the oracle's score rounded to fp32 within 1 ulp (6e-8);
    the dense fp32 MFMA score matrix (dlc_cosine_scores) -> within dlc_cosine_score_error_bound (north_star
    asks 1e-4).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dlc():
    import deeploopcloser_amd as d
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    d.default_engine()          # raises loudly if libdlc_hip.so is missing
    return d


@pytest.fixture(scope="module")
def eng(dlc):
    return dlc.default_engine()


# --------------------------------------------------------------------------- cosine + top-k
def stored(eng, x, dtype):
    return eng.normalize(torch.from_numpy(np.ascontiguousarray(x)).to(eng.device), dtype)


def assert_topk_matches(s, i, q_st, db_st, k, row_offset=0, s64=None):
    """Compare a GPU top-k with the fp64 oracle on the stored values: IDENTICAL indices (north_star), everywhere --
    the GPU decides the order on fp64 scores with the oracle's tie rule.  Scores: fp64 to 1e-12 (summation order),
    fp32 = one rounding of it."""
    from oracle import cosine as ocos
    qn, dbn = q_st.float().cpu().numpy().astype(np.float64), db_st.float().cpu().numpy().astype(np.float64)
    es, ei = ocos.cosine_topk(qn, dbn, k, row_offset=row_offset)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    kk = es.shape[1]
    assert np.all(i[:, kk:] == -1) and np.all(np.isneginf(s[:, kk:]))
    s, i = s[:, :kk], i[:, :kk]
    assert np.array_equal(i, ei)
    assert np.abs(s - es).max() < 1.2e-7
    if s64 is not None:
        assert np.abs(s64.cpu().numpy()[:, :kk] - es).max() < 1e-12


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("nq,n,d,k", [(5, 1000, 64, 10), (256, 5000, 512, 20), (300, 777, 128, 7),
                                      (1, 1, 64, 1), (3, 15, 64, 20), (17, 4097, 192, 128), (64, 20000, 4096, 20)])
def test_cosine_topk_vs_oracle(eng, dtype, nq, n, d, k):
    rng = np.random.RandomState(nq * 31 + n)
    db = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    db_st, q_st = stored(eng, db, dtype), stored(eng, q, dtype)
    s, i = eng.match_topk(q_st, db_st, k, row_offset=1000)
    torch.cuda.synchronize()
    assert_topk_matches(s, i, q_st, db_st, k, row_offset=1000)


@pytest.mark.parametrize("nq,n,d,k", [(5, 1000, 8192, 10), (300, 700, 4160, 7), (1, 1063, 75008, 20), (2, 257, 1088, 3),
                                      (3, 2000, 16384, 128), (32, 5000, 8192, 1)])
def test_cosine_topk_split_k_vs_oracle(eng, nq, n, d, k):
    """Few rows with long descriptors (the reference's own scale: 1063 frames x 75 000) are scored
    split-K: partial tiles + a reducing pass.  Includes ragged last chunks (65 and 17 K tiles)."""
    assert eng.topk_workspace_bytes(nq, n, d, k) > eng.topk_workspace_bytes(nq, n, 64, k)     # split-K partials planned
    rng = np.random.RandomState(n + d)
    db = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = db[n // 3] + 0.2 * q[0]
    db[n - 1] = db[n // 3]                                     # a duplicate in another tile: tie -> lower index
    db_st, q_st = stored(eng, db, "bf16"), stored(eng, q, "bf16")
    s, i = eng.match_topk(q_st, db_st, k, row_offset=7)
    torch.cuda.synchronize()
    assert i[0, 0].item() == n // 3 + 7 and (k < 2 or i[0, 1].item() == n - 1 + 7)
    assert_topk_matches(s, i, q_st, db_st, k, row_offset=7)


def test_cosine_scores_split_k_vs_oracle(eng):
    from oracle import cosine as ocos
    rng = np.random.RandomState(12)
    x = rng.standard_normal((333, 20000)).astype(np.float32)
    st = stored(eng, x, "f16")
    assert eng.lib.dlc_cosine_scores_workspace_bytes(333, 333, st.shape[1]) > 0
    s = eng.cosine_scores(st[:100], st)
    ref = ocos.scores(st[:100].float().cpu().numpy(), st.float().cpu().numpy())
    assert s.shape == (100, 333) and np.abs(s.cpu().numpy() - ref).max() < 2e-5
    s2 = eng.cosine_scores(st[:100], st)
    assert torch.equal(s, s2)                                  # chunk-ordered reduction: bit-reproducible


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("nq,n,d,k", [(1, 1, 64, 1), (1, 100, 4096, 5), (2, 5000, 4096, 20), (3, 70001, 1024, 20),
                                      (4, 3000, 8192, 7), (1, 130, 8192, 128), (4, 999, 192, 20)])
def test_cosine_topk_few_queries_bandwidth_kernel(eng, dtype, nq, n, d, k):
    """q <= 4 (a single frame's query) is scored by the streaming dot-product kernel instead of the
    256-query MFMA tile; same selection + exact re-score after it, same results."""
    rng = np.random.RandomState(n + nq)
    db = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    if n > 40:
        q[0] = db[n // 2] + 0.3 * q[0]
        db[n - 3] = db[n // 2]                                 # duplicate rows: the tie goes to the lower index
    db_st, q_st = stored(eng, db, dtype), stored(eng, q, dtype)
    s, i = eng.match_topk(q_st, db_st, k, row_offset=11)
    torch.cuda.synchronize()
    if n > 40:
        assert i[0, 0].item() == n // 2 + 11 and (k < 2 or i[0, 1].item() == n - 3 + 11)
    assert_topk_matches(s, i, q_st, db_st, k, row_offset=11)


def test_cosine_topk_plan_boundaries_fuzz(eng):
    """Shapes on and around every plan boundary of dlc_cosine_topk (streaming kernel q <= 4 / LDS limit,
    masked query blocks, split-K, multi-workgroup re-score, tile edges), seeded, against the oracle."""
    rng = np.random.RandomState(2026)
    shapes = [(4, 256, 8192, 3), (5, 256, 8192, 3), (4, 257, 8256, 5), (2, 511, 16384, 9), (3, 512, 16448, 1),
              (32, 300, 8192, 20), (33, 300, 8192, 20), (192, 1500, 256, 4), (193, 1500, 256, 4), (256, 255, 64, 128),
              (257, 256, 64, 2), (449, 2049, 128, 6), (1, 32769, 64, 20), (7, 40000, 64, 20), (64, 33000, 1024, 20),
              (512, 4000, 512, 10),
              # small-database plan (<= 16384 rows, > 4 queries): on / around its row and query boundaries
              (5, 16384, 128, 20), (5, 16385, 128, 20), (4, 16384, 128, 20), (64, 16383, 256, 128), (300, 16384, 64, 3),
              (9, 8191, 2048, 20), (1063, 1063, 1024, 20)]
    for _ in range(10):
        shapes.append((int(rng.choice([1, 3, 4, 5, 31, 64, 200, 260])), int(rng.randint(1, 6000)),
                       64 * int(rng.randint(1, 40)), int(rng.choice([1, 2, 20, 77, 128]))))
    for nq, n, d, k in shapes:
        db = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
        if n > 3:
            q[0] = db[n // 2]
            db[n - 1] = db[n // 2]                              # an exact duplicate at the very end
        dtype = "bf16" if (n + nq) % 2 else "f16"
        db_st, q_st = stored(eng, db, dtype), stored(eng, q, dtype)
        s, i = eng.match_topk(q_st, db_st, k, row_offset=3)
        torch.cuda.synchronize()
        if n > 3:
            assert i[0, 0].item() == n // 2 + 3, (nq, n, d, k)
            assert k < 2 or i[0, 1].item() == n - 1 + 3, (nq, n, d, k)
        assert_topk_matches(s, i, q_st, db_st, k, row_offset=3)


def test_cosine_topk_planted_neighbours_exact(eng):
    """Planted-margin data: indices must be IDENTICAL to the oracle's."""
    rng = np.random.RandomState(7)
    n, d, nq, k = 30000, 1024, 256, 20
    db = rng.uniform(0, 1, (n, d)).astype(np.float32)
    db -= db.mean(axis=1, keepdims=True)
    pi = rng.choice(n, nq, replace=False)
    q = db[pi] + 0.12 * rng.standard_normal((nq, d)).astype(np.float32)
    db_st, q_st = stored(eng, db, "bf16"), stored(eng, q, "bf16")
    s, i = eng.match_topk(q_st, db_st, k)
    assert np.array_equal(i[:, 0].cpu().numpy(), pi)           # recall@1 == 1
    assert_topk_matches(s, i, q_st, db_st, k)


def test_cosine_topk_duplicates_break_ties_to_lower_index(eng):
    """Collisions: many identical key-frames (a robot standing still)."""
    rng = np.random.RandomState(3)
    n, d = 6000, 128
    db = rng.standard_normal((n, d)).astype(np.float32)
    dup = np.array([5, 17, 300, 301, 1023, 1024, 2047, 2048, 2049, 3000, 3001, 3002, 3003, 4000, 4500, 4999, 5000,
                    5100, 5200, 5300, 5400, 5500, 5600, 5700, 5800, 5900, 5999])
    db[dup] = db[dup[0]]
    q = np.stack([db[5], db[100]])
    db_st, q_st = stored(eng, db, "bf16"), stored(eng, q, "bf16")
    s, i = eng.match_topk(q_st, db_st, 20)
    assert i[0].cpu().tolist() == sorted(dup.tolist())[:20]
    assert_topk_matches(s, i, q_st, db_st, 20)
    # whole database identical: top-k must be rows 0..k-1
    same = np.repeat(db[:1], 3000, axis=0)
    st = stored(eng, same, "bf16")
    s, i = eng.match_topk(st[:4], st, 33)
    assert i.cpu().tolist() == [list(range(33))] * 4


def test_cosine_topk_crowded_scores_take_the_exhaustive_pass(eng):
    """More near-ties at the k-th place than the selection's slack holds: 300 key-frames that differ from one another by
    ONE bf16 ulp in one element (score differences ~1e-6, below the score pass's error bound tau), spread over 300
    distinct 8-row groups.  The certificate must refuse (status 2 = resolved by the exhaustive pass) and the result must
    still be the fp64 oracle's, index for index -- in the fused call, the two-stage call, the multi-workgroup re-score
    plan (1 query x long rows) and the small-database plan."""
    rng = np.random.RandomState(77)
    for nq, n, d, k, dtype in ((6, 40000, 1024, 20, "bf16"), (3, 40000, 1024, 20, "f16"), (1, 3000, 16384, 10, "bf16"),
                               (8, 9000, 4096, 20, "bf16")):
        db = rng.standard_normal((n, d)).astype(np.float32)
        db_st = stored(eng, db, dtype)
        base = db_st[17].clone()
        where = rng.choice(n // 8, 300, replace=False) * 8 + rng.randint(0, 8, 300)      # one row in each of 300 groups
        bits = db_st.view(torch.int16)
        for j, r in enumerate(where.tolist()):
            bits[r] = base.view(torch.int16)
            bits[r, (37 * j) % d] += 1 if j % 2 else -1                                  # one ulp up / down in one element
        q_st = db_st[torch.tensor([17] + list(range(100, 100 + nq - 1)), device=eng.device)].clone()
        top = eng.match_topk(q_st, db_st, k, row_offset=5, details=True)
        torch.cuda.synchronize()
        assert int(top.status[0]) == 2, (nq, n, d)                 # the crowded query went through the exhaustive pass
        assert set(top.status.cpu().tolist()) <= {0, 2}
        assert_topk_matches(top.scores, top.idx, q_st, db_st, k, row_offset=5, s64=top.scores_f64)
        if nq > 4:                                                 # the two-stage form, both kernel footprints
            for coop in (False, True):
                ws = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
                s2 = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
                i2 = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
                eng.score_groups(q_st, db_st, k, ws)
                eng.select_topk(q_st, db_st, k, ws, s2, i2, row_offset=5, coop=coop)
                assert torch.equal(i2, top.idx) and torch.equal(s2, top.scores)


def _ladder_database(n, d, dtype, seed, g_unit, rungs=40, scale=8.0):
    """Stored rows built on the HOST (what a saved shard or another tool hands over): unit rows, `rungs` copies of row 17
    in `rungs` distinct 8-row groups, copy j lowered by j steps of a few ulps in ONE element -- picked from the binade that
    makes the fp64 scores against row 17 s_0 - j g with g within a factor 2 of g_unit -- then everything times `scale` (a power of two: exact in bf16 / fp16,
    scores times scale^2).  Returns (rows [n, d] as torch CPU tensor of `dtype`, where [rungs], g)."""
    rng = np.random.RandomState(seed)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    x = rng.standard_normal((n, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    st = torch.from_numpy(x).to(tdt)
    bits = st.view(torch.int16)
    mant_bits, step = (7, 1) if dtype == "bf16" else (10, 8)
    base = bits[17].clone()
    # g = step * ulp * |x_e| = step * 2^(-b - mant_bits) * |x_e| with |x_e| in [2^-b, 2^-b+1): the binade b for g_unit, and an
    # element of row 17 in it whose mantissa leaves room for rungs * step ulps below it
    b = int(round((-np.log2(g_unit / (1.5 * step)) - mant_bits) / 2))
    vals = st[17].double().abs()
    mant = (base.to(torch.int32) & ((1 << mant_bits) - 1))
    ok = (vals >= 2.0 ** -b) & (vals < 2.0 ** (1 - b)) & (mant >= rungs * step + (1 << mant_bits) // 8)
    e = int(torch.nonzero(ok)[0])
    where = np.sort(rng.choice(np.arange(3, n // 8), rungs, replace=False)) * 8 + rng.randint(0, 8, rungs)
    where[0] = 17                                            # rung 0 is row 17 itself
    for j, r in enumerate(where.tolist()):
        bits[r] = base
        bits[r, e] = base[e] - j * step                     # magnitude down by j * step ulps (sign bit untouched)
    g = step * 2.0 ** (-b - mant_bits) * float(vals[e])
    return (st.double() * scale).to(tdt), where, g * scale * scale


@pytest.mark.parametrize("n,d,dtype", [(4096, 4096, "bf16"), (40000, 4096, "bf16"), (40000, 4096, "f16")])
def test_cosine_certificate_follows_operand_norms(eng, dlc, n, d, dtype):
    """Rows of norm 8 (queries too: scores x 64, the score pass's fp32 error x 64).  A ladder of 40 near-copies of the query
    row in 40 groups, fp64 scores ~3 tau apart (tau of the shape's plan: 1.6e-5 one-pass, 1.2e-6 split-K): the k-th (20th)
    score clears the best row left behind (the 25th) by ~15 tau -- MORE than the static tau (derived for norms <= 1.005:
    include/dlc.h NORMS), LESS than what the score pass can err by on these operands (tau x 64).  The static tau would certify on margin; the database
    measures its rows' norms (stored=True), the certificate follows them, the query goes through the exhaustive pass
    (status 2) and the list is the fp64 oracle's index for index -- small-database plan (4096 rows), standard plan, fp16;
    one-shot call, two-stream pipeline and the sharded protocol (4 shards, every entry point that takes tau_scale)."""
    from oracle import cosine as ocos
    k, nq = 20, 6
    tau = eng.score_error_bound(nq, n, d, k)
    rows_h, where, g = _ladder_database(n, d, dtype, seed=n + d, g_unit=tau / 20)
    db = dlc.KeyframeDatabase(rows_h, dtype=dtype, stored=True)
    assert db.norm_bound is not None and 8.0 <= float(db.norm_bound) <= 8.1
    q = db.rows[torch.tensor([17, 100, 101, 102, 103, 104], device=eng.device)].clone()
    ts = db.tau_scale(q)
    assert ts.shape == (nq,) and 63.0 < float(ts.min()) and float(ts.max()) < 65.0
    es, ei = ocos.cosine_topk(q.double().cpu().numpy(), rows_h.double().numpy(), k)
    assert ei[0].tolist() == where[:k].tolist()                              # the ladder's first 20 rungs, in order
    gap = es[0, k - 1] - float(q[0].double() @ db.rows[int(where[k + 4])].double())
    assert 2 * tau < gap < 0.5 * tau * float(ts[0])                         # the window the test is about (see above)
    top = db.match_topk(q, k, details=True)
    torch.cuda.synchronize()
    assert int(top.status[0]) == 2 and set(top.status.cpu().tolist()) <= {0, 2}
    assert np.array_equal(top.idx.cpu().numpy(), ei)
    assert np.abs(top.scores_f64.cpu().numpy() - es).max() < 64 * 1e-12
    assert np.abs(top.scores.cpu().numpy() - es).max() < 64 * 1.2e-7
    # what the static tau does with the same operands: certifies the ladder query on its margin (the precondition the
    # header used to state and nothing checked)
    raw = eng.match_topk(q, db.rows, k, details=True)
    assert int(raw.status[0]) == 0
    # unit rows through the same path: scale 1, nothing changes
    unit = dlc.KeyframeDatabase((rows_h.double() / 8).to(rows_h.dtype), dtype=dtype, stored=True)
    assert 1.0 <= float(unit.norm_bound) <= 1.01
    qu = unit.rows[torch.tensor([17, 100], device=eng.device)].clone()
    assert float(unit.tau_scale(qu).max()) <= 1.01
    assert eng.unit_rows(eng.normalize(qu.float(), dtype)) and not eng.unit_rows(qu) and unit.tau_scale(
        eng.normalize(qu.float(), dtype)) is not None                        # the ROWS are foreign, whatever the queries are
    # the mark is outdated by the library's own raw-pointer writes as by torch's: an upload into a normalised tensor (or a
    # view of it) leaves rows of any norm behind
    marked = eng.normalize(qu.float(), dtype)
    assert eng.unit_rows(marked)
    eng.upload(np.full(tuple(marked[:1].shape), 0x4040, dtype=np.int16), out=marked[:1].view(torch.int16))   # 3.0 / 2.125 everywhere
    torch.cuda.synchronize()
    assert float(marked[0].float().norm()) > 10
    assert not eng.unit_rows(marked)
    own = dlc.KeyframeDatabase(rows_h.float(), dtype=dtype)                  # normalised here: trusted, no scale
    assert own.norm_bound is None and own.tau_scale(own.prepare_queries(qu.float())) is None
    # the two-stream pipeline on one GPU
    pipe = dlc.MatchPipeline(db, k)
    s2, i2 = pipe.result(pipe.submit(q))
    assert torch.equal(i2, top.idx) and torch.equal(s2, top.scores)
    # the sharded protocol with 4 shards: select groups, exchange maxima, filtered re-score, certifying merge, exhaustive round
    parts, kg = 4, eng.groups_per_query(k)
    tau_any = eng.score_error_bound_any_plan(d)
    assert tau_any >= tau
    ids, mx, wss, shards = [], [], [], [dlc.shard_bounds(n, parts, r) for r in range(parts)]
    for lo, hi in shards:
        ws = torch.empty(eng.topk_workspace_bytes(nq, hi - lo, d, k), dtype=torch.uint8, device=eng.device)
        gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
        gm = torch.empty((nq, kg + 1), dtype=torch.float32, device=eng.device)
        eng.score_groups(q, db.rows[lo:hi], k, ws)
        eng.select_groups(q, db.rows[lo:hi], k, ws, gi, gm)
        ids.append(gi), mx.append(gm), wss.append(ws)
    all_max = torch.stack(mx)
    gathered = torch.empty((parts, nq * k * 16), dtype=torch.uint8, device=eng.device)
    bound = torch.empty((nq,), dtype=torch.float32, device=eng.device)
    for r, (lo, hi) in enumerate(shards):
        eng.rescore_topk(q, db.rows[lo:hi], k, ids[r], mx[r], gathered[r, nq * k * 8:].view(torch.float64).view(nq, k),
                         gathered[r, :nq * k * 8].view(torch.int64).view(nq, k), bound=bound, all_max=all_max, row_offset=lo,
                         tau_scale=ts)
    o_s = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
    o_i = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
    o_64 = torch.empty((nq, k), dtype=torch.float64, device=eng.device)
    status = torch.full((nq,), -1, dtype=torch.int32, device=eng.device)
    eng.topk_merge_packed(gathered, nq, k, out=(o_s, o_i), bound=bound, tau=tau_any, scores_f64=o_64, status=status, tau_scale=ts)
    assert int(status[0]) == 1 and torch.equal(status == 0, o_64[:, k - 1] > bound.double() + tau_any * ts.double())
    static = torch.full((nq,), -1, dtype=torch.int32, device=eng.device)
    eng.topk_merge_packed(gathered, nq, k, out=(o_s, o_i), bound=bound, tau=tau_any, scores_f64=o_64, status=static)
    assert int(static[0]) == 0                                               # (again: the static tau's verdict)
    lower = o_64[:, k - 1].contiguous()
    for r, (lo, hi) in enumerate(shards):
        st = status.clone()
        eng.exhaustive_topk(q, db.rows[lo:hi], k, wss[r], lower, tau_any, st, gathered[r, nq * k * 8:].view(torch.float64).view(nq, k),
                            gathered[r, :nq * k * 8].view(torch.int64).view(nq, k), row_offset=lo, tau_scale=ts)
        assert int(st[0]) == 2
    eng.topk_merge_packed(gathered, nq, k, out=(o_s, o_i), scores_f64=o_64)
    assert torch.equal(o_i, top.idx) and torch.equal(o_64, top.scores_f64)
    # a non-finite element: the scale is +inf, nothing certifies, the exhaustive pass decides (finite queries still exact)
    poisoned = db.rows.clone()
    poisoned[5, 3] = float("inf")
    pdb = dlc.KeyframeDatabase(poisoned, dtype=dtype, stored=True)
    assert bool(torch.isinf(pdb.norm_bound).all()) and bool(torch.isinf(pdb.tau_scale(q)).all())


def test_pipeline_refuses_to_overwrite_an_unverified_batch(eng, dlc):
    """MatchPipeline.submit on a slot whose batch was never fetched: dropping a certified batch loses nothing (counted in
    dropped_batches); a batch whose sharded merge did NOT certify has its exhaustive round pending inside result() -- submit
    raises instead of overwriting the unverified lists (it used to count and carry on).  One process, the
    sharded branch switched on with a second, empty 'rank' (as bench.py's emulation does)."""
    rng = np.random.RandomState(5)
    n, d, k = 30000, 256, 8
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[rng.choice(n, 200, replace=False)] = x[3]                      # 201 copies of row 3: more ties than any selection holds
    db = dlc.KeyframeDatabase(x, dtype="bf16")
    kg = eng.groups_per_query(k)

    def lonely_all_gather(out_t, inp, group=None):                   # rank 1 has nothing: -inf maxima, empty lists
        o = out_t.view(2, -1)
        o[0].copy_(inp.reshape(-1))
        if inp.dtype == torch.float32:
            o[1].fill_(float("-inf"))
        else:
            nq = inp.numel() // (k * 16)
            o[1][:nq * k * 8].view(torch.int64).fill_(-1)
            o[1][nq * k * 8:].view(torch.float64).fill_(float("-inf"))

    def pipe_of():
        p = dlc.MatchPipeline(db, k, depth=1)
        p.world = 2
        p.all_gather = lonely_all_gather
        return p
    clean = db.prepare_queries(x[100:104])
    crowded = db.prepare_queries(x[[3, 100]])
    p = pipe_of()
    p.submit(clean); p.submit(clean)                                 # certified, never fetched: dropped, counted
    assert p.dropped_batches == 1
    s_, i_ = p.result(1)
    assert i_[:, 0].cpu().tolist() == [100, 101, 102, 103]
    p = pipe_of()
    t = p.submit(crowded)
    with pytest.raises(RuntimeError, match="never fetched and its merge did not certify 1 query"):
        p.submit(crowded)
    s_, i_ = p.result(t)                                             # the round still runs where it belongs
    assert p.resolved_batches == 1
    want = eng.match_topk(crowded, db.rows, k)
    assert torch.equal(i_, want[1]) and torch.equal(s_, want[0])


def test_any_plan_error_bound_covers_every_plan(eng):
    """dlc_cosine_score_error_bound_any_plan(d) -- what a sharded merge certifies with -- is at least the tau of whatever
    plan a shard's shape picks: the bandwidth kernel (q <= 4), the small-database plan, split-K, the one-pass MFMA plan."""
    for d in (64, 128, 256, 1024, 4096, 16384, 75008):
        any_plan = eng.score_error_bound_any_plan(d)
        for nq in (1, 2, 4, 5, 32, 256):
            for n in (8, 1000, 16384, 16385, 125_000, 1_000_000, 1 << 30):
                assert eng.score_error_bound(nq, n, d, 20) <= any_plan, (nq, n, d)


def test_score_error_bound_holds(eng):
    """dlc_cosine_score_error_bound is what the certificate of every selection rests on: |fp32 score of the score pass -
    fp64 score| <= tau.  Checked on every element of dense MFMA score matrices (one pass and split-K) for random,
    all-positive (partial sums grow monotonically: the worst case of the bound's model) and cancelling data, and for the
    q <= 4 bandwidth kernel through its group maxima; the worst observed fraction of tau is printed."""
    rng = np.random.RandomState(5)
    worst = 0.0
    for nq, n, d, dtype in ((64, 700, 64, "bf16"), (64, 700, 4096, "bf16"), (64, 700, 4096, "f16"), (32, 300, 16384, "bf16"),
                            (16, 520, 75008, "bf16")):
        for kind in ("normal", "positive", "cancel"):
            x = rng.standard_normal((n, d)).astype(np.float32)
            y = rng.standard_normal((nq, d)).astype(np.float32)
            if kind == "positive":
                x, y = np.abs(x), np.abs(y)
            elif kind == "cancel":                                  # +a, -a pairs: products cancel, partial sums do not
                x[:, 1::2] = -x[:, 0::2]
                y[:, 1::2] = y[:, 0::2]
            db_st, q_st = stored(eng, x, dtype), stored(eng, y, dtype)
            tau = eng.score_error_bound(nq, n, db_st.shape[1], 20)
            s = eng.cosine_scores(q_st, db_st).double()
            ref = q_st.double() @ db_st.double().T
            err = float((s - ref).abs().max())
            worst = max(worst, err / tau)
            assert err <= tau, (nq, n, d, dtype, kind, err, tau)
    # q <= 4: v_dot2 chains; the group maxima of select_groups are the kernel's fp32 scores
    for nq, n, d, dtype in ((1, 3000, 4096, "bf16"), (4, 3000, 8192, "f16"), (2, 3000, 64, "bf16")):
        x = np.abs(rng.standard_normal((n, d))).astype(np.float32)
        db_st = stored(eng, x, dtype)
        q_st = stored(eng, np.abs(rng.standard_normal((nq, d))).astype(np.float32), dtype)
        k = 20
        kg = eng.groups_per_query(k)
        tau = eng.score_error_bound(nq, n, db_st.shape[1], k)
        ws = torch.empty(eng.topk_workspace_bytes(nq, n, db_st.shape[1], k), dtype=torch.uint8, device=eng.device)
        gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
        gm = torch.empty((nq, kg + 1), dtype=torch.float32, device=eng.device)
        eng.score_groups(q_st, db_st, k, ws)
        eng.select_groups(q_st, db_st, k, ws, gi, gm)
        ref = (q_st.double() @ db_st.double().T)[:, :n // 8 * 8].reshape(nq, n // 8, 8).max(dim=2).values   # exact group maxima
        got = gm[:, :kg].double()
        want = torch.gather(ref, 1, gi.long().clamp(min=0))
        err = float((got - want).abs().max())
        worst = max(worst, err / tau)
        assert err <= tau, (nq, n, d, dtype, err, tau)
        # and the listed groups are the best ones up to tau: nothing left behind beats the last listed maximum by more
        assert float((ref.max(dim=1).values - got[:, 0]).abs().max()) <= tau
    print("score error bound: worst observed error = %.3f of tau" % worst)


def test_cosine_topk_row_stride_and_padding(eng):
    """d not a multiple of 64 is zero-padded by normalize(); a strided view works."""
    rng = np.random.RandomState(11)
    db = rng.standard_normal((2500, 100)).astype(np.float32)
    q = rng.standard_normal((9, 100)).astype(np.float32)
    db_st, q_st = stored(eng, db, "bf16"), stored(eng, q, "bf16")
    assert db_st.shape[1] == 128 and torch.all(db_st[:, 100:] == 0)
    big = torch.zeros((2500, 256), dtype=torch.bfloat16, device=eng.device)
    big[:, :128] = db_st
    s, i = eng.match_topk(q_st, big[:, :128], 10)
    assert_topk_matches(s, i, q_st, db_st, 10)
    s, i = eng.match_topk(q_st[:2], big[:, :128], 10)           # q <= 4: the streaming kernel, same strided rows
    assert_topk_matches(s, i, q_st[:2], db_st, 10)
    qbig = torch.zeros((9, 192), dtype=torch.bfloat16, device=eng.device)
    qbig[:, :128] = q_st
    s, i = eng.match_topk(qbig[:3, :128], big[:, :128], 10)     # strided queries as well
    assert_topk_matches(s, i, q_st[:3], db_st, 10)


def test_cosine_scores_dense_vs_oracle(eng):
    from oracle import cosine as ocos
    rng = np.random.RandomState(2)
    x = rng.standard_normal((333, 320)).astype(np.float32)
    st = stored(eng, x, "bf16")
    s = eng.cosine_scores(st[:70], st).cpu().numpy()
    ref = ocos.scores(st[:70].float().cpu().numpy(), st.float().cpu().numpy())
    assert s.shape == (70, 333) and np.abs(s - ref).max() < 2e-5


def test_topk_merge_vs_oracle(eng):
    from oracle import cosine as ocos
    rng = np.random.RandomState(4)
    parts, nq, k = 8, 50, 20
    s = rng.uniform(-1, 1, (parts, nq, k))
    s[3, :, 5] = s[1, :, 2]
    s[4, :, 0] = s[2, :, 7] + 2.0 ** -45                       # closer than the 2^-40 key: a tie, the lower row wins
    s[5, :, 1] = np.float32(s[6, :, 3]).astype(np.float64)     # equal after rounding to fp32, not in fp64: NOT a tie
    i = rng.permutation(parts * nq * k).reshape(parts, nq, k).astype(np.int64)
    i[7, :, -3:] = -1
    r = eng.topk_merge(torch.from_numpy(s).to(eng.device), torch.from_numpy(i).to(eng.device), details=True)
    cs = np.transpose(s, (1, 0, 2)).reshape(nq, parts * k)
    ci = np.transpose(i, (1, 0, 2)).reshape(nq, parts * k)
    cs = np.where(ci < 0, -np.inf, cs)
    es, ei = ocos.merge_topk(cs, np.where(ci < 0, np.iinfo(np.int64).max, ci), k)
    assert np.array_equal(r.idx.cpu().numpy(), ei) and np.array_equal(r.scores_f64.cpu().numpy(), es)
    assert np.array_equal(r.scores.cpu().numpy(), es.astype(np.float32))


@pytest.mark.parametrize("n", [80000, 40000, 12000])
def test_sharded_equals_unsharded(eng, dlc, n):
    """Size-independent property: 4 row shards + merge == one shard, BIT FOR BIT (indices, fp64 and fp32 scores),
    whatever plans the shards and the whole database take (80 000: all gather their groups' rows; 12 000: all pick
    rows off the fp32 score matrix; 40 000: the 10 000-row shards do the latter, the whole the former)."""
    rng = np.random.RandomState(9)
    d, nq, k = 256, 128, 20
    db_st = stored(eng, rng.standard_normal((n, d)).astype(np.float32), "bf16")
    q_st = stored(eng, rng.standard_normal((nq, d)).astype(np.float32), "bf16")
    r0 = eng.match_topk(q_st, db_st, k, details=True)
    ps, pi = [], []
    for r in range(4):
        lo, hi = dlc.shard_bounds(n, 4, r)
        t = eng.match_topk(q_st, db_st[lo:hi], k, row_offset=lo, details=True)
        ps.append(t.scores_f64.clone()), pi.append(t.idx.clone())
    m = eng.topk_merge(torch.stack(ps), torch.stack(pi), details=True)
    assert torch.equal(m.idx, r0.idx) and torch.equal(m.scores_f64, r0.scores_f64) and torch.equal(m.scores, r0.scores)


def test_match_errors(eng):
    q = torch.zeros((4, 64), dtype=torch.bfloat16, device=eng.device)
    with pytest.raises(ValueError):
        eng.match_topk(q, q, 0)
    with pytest.raises(ValueError):
        eng.match_topk(q, q, 129)
    with pytest.raises(ValueError):
        eng.match_topk(q[:, :32], q[:, :32], 2)                     # d not a multiple of 64
    with pytest.raises(ValueError):
        eng.match_topk(q, q.to(torch.float16), 2)
    with pytest.raises(ValueError):
        eng.normalize(torch.zeros((2, 8), dtype=torch.int32, device=eng.device))


def test_normalize_rows(eng):
    rng = np.random.RandomState(1)
    x = rng.standard_normal((37, 100))
    for center in (False, True):
        got = eng.normalize(torch.from_numpy(x).to(eng.device), "bf16", center).float().cpu().numpy()
        y = x - x.mean(1, keepdims=True) if center else x
        ref = y / np.linalg.norm(y, axis=1, keepdims=True)
        ref_bf = torch.from_numpy(ref.astype(np.float32)).to(torch.bfloat16).float().numpy()
        assert np.abs(got[:, :100] - ref_bf).max() <= 2.0 ** -8 * np.abs(ref).max() and np.all(got[:, 100:] == 0)
        assert (got[:, :100] == ref_bf).mean() > 0.999


# --------------------------------------------------------------------------- dense layers / SDAV
@pytest.mark.parametrize("m,n,k", [(37, 53, 29), (128, 128, 16), (600, 2500, 1681), (1, 1, 1), (130, 384, 3456)])
@pytest.mark.parametrize("blayout", [0, 1])
def test_gemm_bias_act_f64(eng, m, n, k, blayout):
    from oracle import tensor_ops
    rng = np.random.RandomState(m + n + k)
    a = rng.standard_normal((m, k))
    b = rng.standard_normal((k, n)) / np.sqrt(k)
    bias = rng.standard_normal(n)
    bt = torch.from_numpy(np.ascontiguousarray(b if blayout == 0 else b.T)).to(eng.device)
    for act, fn in ((0, lambda z: z), (1, tensor_ops.sigmoid), (2, lambda z: np.maximum(z, 0))):
        got = eng.gemm_bias_act(torch.from_numpy(a).to(eng.device), bt, torch.from_numpy(bias).to(eng.device),
                                act=act, blayout=blayout).cpu().numpy()
        ref = fn(a @ b + bias)
        assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m,n,k,blayout", [(30, 2500, 2500, "kn"), (130, 384, 3456, "kn"), (60, 300, 1681, "nk"), (1, 129, 4000, "kn")])
def test_gemm_split_k_latency_mode(eng, m, n, k, blayout):
    """Few rows x long K (one frame's SDAV layer / conv3-5) run split-K through the engine's scratch:
    same result as one pass up to the summation order, bit-reproducible, 1e-10 from NumPy."""
    from deeploopcloser_amd import _lib as L
    rng = np.random.RandomState(m + n)
    a = rng.standard_normal((m, k)) / np.sqrt(k)
    b = rng.standard_normal((k, n)) if blayout == "kn" else rng.standard_normal((n, k))
    bias = rng.standard_normal(n)
    ref = 1.0 / (1.0 + np.exp(-(a @ (b if blayout == "kn" else b.T) + bias)))
    ta, tb, tbias = (torch.from_numpy(v).to(eng.device) for v in (a, b, bias))
    lay = L.DLC_B_KN if blayout == "kn" else L.DLC_B_NK
    one_pass = eng.gemm_bias_act(ta, tb, tbias, act=L.DLC_ACT_SIGMOID, blayout=lay)
    eng.set_scratch()
    try:
        split = eng.gemm_bias_act(ta, tb, tbias, act=L.DLC_ACT_SIGMOID, blayout=lay)
        again = eng.gemm_bias_act(ta, tb, tbias, act=L.DLC_ACT_SIGMOID, blayout=lay)
    finally:
        eng.set_scratch(0)
    assert torch.equal(split, again)
    assert np.abs(split.cpu().numpy() - ref).max() < 1e-10
    assert np.abs(one_pass.cpu().numpy() - ref).max() < 1e-10
    assert (split - one_pass).abs().max().item() < 1e-12


@pytest.mark.parametrize("m,n,k,blayout", [(300, 2500, 2500, "kn"), (300, 1681, 2500, "nk"), (300, 2501, 2500, "nk"), (77, 130, 4096, "kn"),
                                           (200, 256, 8190, "nk"), (1, 2500, 2502, "kn"), (255, 1000, 644, "kn"), (64, 128, 130, "kn"),
                                           (320, 1024, 1682, "nk"), (129, 98, 5000, "kn")])
def test_gemm_dma_split_k_shapes(eng, m, n, k, blayout):
    """Few 64-row tiles and a long even K: split-K on the LDS-DMA kernel's 64-row tiles (r06: the training step's 300-row
    products, an encode of a few frames) -- chunk tails (K % 16, K % chunk), an odd N for [N,K] operands, a single row, a K too
    short to split (one pass), every activation: 1e-11 from NumPy's fp64 product, the same bits run after run, and within the
    summation order of the one-pass result."""
    from deeploopcloser_amd import _lib as L
    from oracle import tensor_ops
    rng = np.random.RandomState(m * 7 + n + k)
    a = rng.standard_normal((m, k)) / np.sqrt(k)
    b = rng.standard_normal((k, n)) if blayout == "kn" else rng.standard_normal((n, k))
    bias = rng.standard_normal(n)
    z = a @ (b if blayout == "kn" else b.T) + bias
    ta, tb, tbias = (torch.from_numpy(np.ascontiguousarray(v)).to(eng.device) for v in (a, b, bias))
    lay = L.DLC_B_KN if blayout == "kn" else L.DLC_B_NK
    for act, fn in ((L.DLC_ACT_NONE, lambda v: v), (L.DLC_ACT_SIGMOID, tensor_ops.sigmoid), (L.DLC_ACT_RELU, lambda v: np.maximum(v, 0))):
        one_pass = eng.gemm_bias_act(ta, tb, tbias, act=act, blayout=lay)
        with eng.latency_mode():
            split = eng.gemm_bias_act(ta, tb, tbias, act=act, blayout=lay)
            again = eng.gemm_bias_act(ta, tb, tbias, act=act, blayout=lay)
            nobias = eng.gemm_bias_act(ta, tb, None, act=L.DLC_ACT_NONE, blayout=lay)
        ref = fn(z)
        scale = max(1.0, np.abs(ref).max())
        assert torch.equal(split, again)
        assert np.abs(split.cpu().numpy() - ref).max() < 1e-11 * scale
        assert np.abs(one_pass.cpu().numpy() - ref).max() < 1e-11 * scale
        assert np.abs(nobias.cpu().numpy() - (z - bias)).max() < 1e-11 * scale


def test_conv_and_gemm_fuzz_in_both_modes(eng):
    """Seeded random conv geometries (C % 8 == 0) and GEMM shapes, one-pass and split-K (scratch on):
    both within 1e-10 of the oracle; the split start (ky, kx, c) of every K chunk is exercised."""
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(77)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    cases = []
    for _ in range(8):
        kh = int(rng.choice([1, 3, 5]))
        cases.append((int(rng.randint(kh, 14)), int(rng.randint(kh, 14)), 8 * int(rng.randint(1, 40)), kh,
                      int(rng.randint(1, 200)), int(rng.choice([1, 2])), str(rng.choice(["SAME", "VALID"])), bool(rng.randint(2))))
    for mode in (False, True):
        eng.set_scratch(1 << 26 if mode else 0)
        try:
            for (h, w, c, kh, cout, stride, pad, relu) in cases:
                x = rng.standard_normal((2, h, w, c))
                wk = rng.standard_normal((kh, kh, c, cout)) / np.sqrt(kh * kh * c)
                b = rng.standard_normal(cout)
                oh, ph = ocnn._out_size(h, kh, stride, pad)
                ow, pw = ocnn._out_size(w, kh, stride, pad)
                got = eng.conv2d(dev(x), dev(wk.reshape(-1, cout)), dev(b), kh, kh, stride, ph, pw, oh, ow, 2 if relu else 0)
                ref = ocnn.conv2d_nhwc(x, wk, b, stride, pad, relu)
                assert np.abs(got.cpu().numpy() - ref).max() < 1e-10, (mode, h, w, c, kh, cout, stride, pad)
            for _ in range(6):
                m, n, k = int(rng.randint(1, 300)), int(rng.randint(1, 300)), int(rng.randint(1, 5000))
                a, bm, bias = rng.standard_normal((m, k)) / np.sqrt(k), rng.standard_normal((k, n)), rng.standard_normal(n)
                got = eng.gemm_bias_act(dev(a), dev(bm), dev(bias), act=1)
                assert np.abs(got.cpu().numpy() - 1.0 / (1.0 + np.exp(-(a @ bm + bias)))).max() < 1e-10, (mode, m, n, k)
        finally:
            eng.set_scratch(0)


def test_latency_mode_encoders_vs_oracle(dlc, eng):
    """Single-frame SDAV and CnnVtl encodes with the split-K scratch on: same oracle tolerances."""
    from oracle import sdav as osdav, cnn_vtl as ocnn
    rng = np.random.RandomState(21)
    x = rng.uniform(0, 1, (1, 30, 1681))
    frame = rng.randint(0, 256, (1, 192, 240, 3)).astype(np.float64)
    net = dlc.SDAV(seed=5)
    cnn = dlc.CnnVtl(input_shape=[1, 192, 240, 3], seed=3, mask_seed=4)
    eng.set_scratch()
    try:
        h = net.transform(x)
        d = cnn.transform(frame)
    finally:
        eng.set_scratch(0)
    ws, bs = net.get_weights()
    assert np.abs(h - osdav.transform(x, ws, bs)).max() < 1e-10
    cw, cb = ocnn.init_weights(3)
    ref = ocnn.transform(frame, cw, cb, ocnn.column_indices(cnn.layer_sizes, 99.59, seed=4))
    assert d.dtype == np.int8 and np.array_equal(d, ref)          # every byte (a value within ~1e-11 of an integer
    # could truncate differently under another summation order: none does for this seed)


def test_gemm_bias_act_f32(eng):
    rng = np.random.RandomState(0)
    a = rng.standard_normal((300, 777)).astype(np.float32)
    b = (rng.standard_normal((777, 200)) / 28).astype(np.float32)
    got = eng.gemm_bias_act(torch.from_numpy(a).to(eng.device), torch.from_numpy(b).to(eng.device), None, act=1)
    ref = 1 / (1 + np.exp(-(a.astype(np.float64) @ b.astype(np.float64))))
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6


@pytest.mark.parametrize("scale", ["reference", "fan_in"])
def test_sdav_transform_vs_oracle(dlc, scale):
    """SDAV.transform (SDAV.py:293-302), both weight regimes of SURVEY section 7 'hard parts'."""
    from oracle import sdav as osdav
    rng = np.random.RandomState(5)
    net = dlc.SDAV(seed=11, weight_scale=scale)
    x = rng.uniform(0, 1, size=(7, 30, 1681))
    h = net.transform(x)
    ws, bs = net.get_weights()
    ow, ob = osdav.init_weights(11, scale=scale)
    assert all(np.array_equal(a, b) for a, b in zip(ws, ow))
    ref = osdav.transform(x, ws, bs)
    assert h.shape == (210, 2500) and h.dtype == np.float64
    assert np.abs(h - ref).max() < 1e-10
    l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert l2.max() < 1e-10                                          # north_star: descriptor L2 within 1e-4


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("f16x2", 1e-4)])
def test_sdav_other_widths_vs_oracle(dlc, dtype, tol):
    """SDAV(hidden_units=[...]): the same chain with other layer widths (non-reference: SDAV.py:31-32 fixes 5 x 2500) --
    a 4096-wide last layer for north_star's 4096-d descriptors, and a ragged stack -- against the oracle's forward."""
    from oracle import sdav as osdav
    rng = np.random.RandomState(21)
    for hu in ([2500, 2500, 2500, 2500, 4096], [300, 77, 1000]):
        net = dlc.SDAV(seed=5, dtype=dtype, weight_scale="fan_in", hidden_units=hu)
        assert net.hidden_units == hu and net.get_layer_input_shape(len(hu) - 1) == [30, hu[-2]]
        x = rng.uniform(0, 1, size=(4, 30, 1681))
        h = net.transform(x)
        ws, bs = net.get_weights()
        assert [w.shape for w in ws] == [(k, n) for k, n in zip([1681] + hu[:-1], hu)]
        ref = osdav.transform(x, ws, bs)
        assert h.shape == (120, hu[-1]) and h.dtype == np.float64
        l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert l2.max() < tol
    with pytest.raises(ValueError):
        dlc.SDAV(hidden_units=[])
    with pytest.raises(ValueError):
        dlc.SDAV(hidden_units=[2500, 0])


def test_sdav_transform_fp32_mode_within_north_star_tolerance(dlc):
    from oracle import sdav as osdav
    rng = np.random.RandomState(6)
    net = dlc.SDAV(seed=12, dtype="float32", weight_scale="fan_in")
    x = rng.uniform(0, 1, size=(3, 30, 1681))
    h = net.transform(x)
    ws, bs = net.get_weights()
    ref = osdav.transform(x, [w.astype(np.float64) for w in ws], [b.astype(np.float64) for b in bs])
    l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert l2.max() < 1e-4


@pytest.mark.parametrize("scale", ["reference", "fan_in"])
def test_sdav_transform_split_mode_within_north_star_tolerance(dlc, scale):
    """SDAV(dtype="f16x2"): the tolerance mode on the 16-bit MFMA (three fp16 products of two-piece splits per layer,
    csrc/gemm_split_f16.hip) against the fp64 oracle -- north_star's "descriptor L2 within 1e-4" -- on random frames and on
    the 20 real frames of datasets/test, in BOTH weight regimes: the reference's N(0,1) initialiser (the hard one: errors
    are amplified through five saturating layers) and 1/sqrt(fan_in).  One frame alone == that frame inside a batch."""
    import config1_common as c1
    from oracle import sdav as osdav
    rng = np.random.RandomState(7)
    net = dlc.SDAV(seed=12, dtype="f16x2", weight_scale=scale)
    ws, bs = net.get_weights()
    assert ws[0].dtype == np.float64
    bs = [0.1 * rng.standard_normal(b.shape) for b in bs]              # non-zero biases take part too
    net.set_weights(ws, bs)
    worst = 0.0
    for name, x in (("random", rng.uniform(0, 1, size=(9, 30, 1681))), ("real", c1.oracle_patches(c1.frame_paths()))):
        h = net.transform(x)
        ref = osdav.transform(x, ws, bs)
        assert h.shape == ref.shape and h.dtype == np.float64
        l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
        print("SDAV f16x2, %s weights, %s frames: descriptor relative L2 max %.3g median %.3g, max abs err %.3g"
              % (scale, name, l2.max(), np.median(l2), np.abs(h - ref).max()))
        assert l2.max() < 1e-4
        worst = max(worst, l2.max())
        one = net.transform(x[3:4])
        assert np.array_equal(one, h[90:120])                          # batch invariance (rows are independent)
    assert worst < (4e-5 if scale == "reference" else 1e-6)            # what the form delivers (2.1e-5 / 1.8e-7), with margin


@pytest.mark.parametrize("rows,dims", [(1, (37, 21)), (300, (37, 21, 19, 23)), (257, (1681, 250, 2)), (513, (64, 300, 257))])
def test_sdav_split_mode_ragged_shapes(dlc, rows, dims):
    """dlc_sdav_encode_split on shapes the drop-in class never asks for: odd widths (the fp64 output's pitch is odd: no
    16-byte stores), widths that end inside a 16-column block, a 64-column half and a 256-column tile, a single layer, one
    row, rows that end just behind a 256-row tile; no biases for one layer.  Against the fp64 chain in NumPy: relative L2
    of every row within 1e-4, and every row the same when encoded alone (rows are independent)."""
    eng = dlc.default_engine()
    rng = np.random.RandomState(rows + len(dims))
    n_layers = len(dims) - 1
    ws = [rng.standard_normal((dims[l], dims[l + 1])) / np.sqrt(dims[l]) for l in range(n_layers)]
    bs = [0.2 * rng.standard_normal(dims[l + 1]) if l != 1 else None for l in range(n_layers)]
    x = rng.uniform(0, 1, size=(rows, dims[0]))
    ref = x
    for w, b in zip(ws, bs):
        ref = 1.0 / (1.0 + np.exp(-(ref @ w + (0.0 if b is None else b))))
    tw = [torch.from_numpy(w).to(eng.device) for w in ws]
    tb = [None if b is None else torch.from_numpy(b).to(eng.device) for b in bs]
    panels = eng.sdav_split_panels(tw)
    tx = torch.from_numpy(x).to(eng.device)
    h = eng.sdav_encode_split(tx, list(dims), panels, tb).cpu().numpy()
    assert h.shape == ref.shape
    l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert l2.max() < 1e-4, l2.max()
    for r in sorted({0, rows // 2, rows - 1}):
        one = eng.sdav_encode_split(tx[r:r + 1], list(dims), panels, tb).cpu().numpy()
        assert np.array_equal(one[0], h[r])


@pytest.mark.parametrize("how", ["train_step", "train_steps_graph", "fit"])
def test_sdav_split_mode_sees_trained_weights(dlc, how):
    """transform -> train -> transform in the tolerance mode (SDAV.py:232-240,293-302: transform always sees the current
    variables).  The training kernels update the weights in place through raw pointers -- neither data_ptr() nor torch's
    version counter moves -- so the fp16 panels are keyed on the network's own weights generation (the cache
    used to return descriptors of the PRE-training weights).  After training: a fresh network given the trained weights
    returns the same descriptors bit for bit, they differ from the ones before, and they are within 1e-4 relative L2 of
    the fp64 oracle on the trained weights."""
    from oracle import sdav as osdav
    rng = np.random.RandomState(17)
    net = dlc.SDAV(seed=13, dtype="f16x2", weight_scale="fan_in")
    x = rng.uniform(0, 1, size=(6, 30, 1681))
    xb = rng.uniform(0, 1, size=(4, 30, 1681))
    h0 = net.transform(x)
    if how == "train_step":
        for _ in range(3):
            net.train_step(0, xb)
    elif how == "train_steps_graph":
        with net.engine.latency_mode():
            net.train_steps(1, xb, 5)                                   # one eager step, then graph replays
            h_mid = net.transform(x)
            net.train_steps(1, xb, 4)                                   # replays only: the cached graph
        assert not np.array_equal(h_mid, net.transform(x))
    else:
        net.epochs = 3
        net.fit(xb)
    h1 = net.transform(x)
    ws, bs = net.get_weights()
    fresh = dlc.SDAV(seed=99, dtype="f16x2")
    fresh.set_weights(ws, bs)
    assert np.array_equal(fresh.transform(x), h1)
    assert not np.array_equal(h1, h0)
    ref = osdav.transform(x, ws, bs)
    l2 = np.linalg.norm(h1 - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert l2.max() < 1e-4
    l2_stale = np.linalg.norm(h0 - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert l2_stale.max() > 10 * l2.max()                               # the test can tell stale from fresh
    # set_weights on the same object: new tensors that the caching allocator may place where the old ones lived
    net.set_weights([w * 0.5 for w in ws], bs)
    ref2 = osdav.transform(x, [w * 0.5 for w in ws], bs)
    h2 = net.transform(x)
    assert (np.linalg.norm(h2 - ref2, axis=1) / np.linalg.norm(ref2, axis=1)).max() < 1e-4


def test_sdav_encode_split_odd_shapes(eng):
    """dlc_sdav_encode_split on widths that are no multiple of anything (K tails, ragged last tiles, one layer, a single
    row) against a float64 torch chain: relative error of every output below 1e-5; re-preparing after a weight change."""
    g = torch.Generator(device=eng.device); g.manual_seed(3)
    for rows, dims in ((1, [5, 3]), (30, [1681, 2500, 77]), (257, [100, 300, 513, 64]), (1000, [64, 64]), (513, [333, 2, 1, 9])):
        ws = [torch.randn((dims[l], dims[l + 1]), generator=g, device=eng.device, dtype=torch.float64) / np.sqrt(dims[l])
              for l in range(len(dims) - 1)]
        bs = [0.3 * torch.randn((dims[l + 1],), generator=g, device=eng.device, dtype=torch.float64) for l in range(len(dims) - 1)]
        x = torch.rand((rows, dims[0]), generator=g, device=eng.device, dtype=torch.float64)
        for rep in range(2):
            panels = eng.sdav_split_panels(ws)
            got = eng.sdav_encode_split(x, dims, panels, bs)
            ref = x
            for w, b in zip(ws, bs):
                ref = torch.sigmoid(ref @ w + b)
            assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-5, (rows, dims)
            ws[0] = ws[0] * 3.0                                          # other weights (another scale exponent): prepare again
    with pytest.raises(ValueError):
        eng.sdav_encode_split(x.float(), dims, panels, bs)


def test_sdav_surface(dlc):
    net = dlc.SDAV()
    assert net.input_shape == [30, 1681] and net.hidden_units == [2500] * 5 and net.default_batch_size == 10
    assert net.get_layers_input_shapes() == [[30, 2500]] * 5
    assert net.transform(np.zeros((0, 30, 1681))).shape == (0, 2500)
    with pytest.raises(ValueError):
        net.transform(np.zeros((2, 30, 100)))
    with pytest.raises(ValueError):
        net.fit(np.zeros((1, 30, 1681)))           # one frame: the consecutive-frame term needs two


def test_da_transform_vs_oracle(dlc):
    from oracle import sdav as osdav
    rng = np.random.RandomState(8)
    da = dlc.DA([30, 1681], 2500, seed=4)
    x = rng.uniform(0, 1, size=(30, 1681))
    w = da._w0.cpu().numpy()
    assert np.abs(da.transform(x) - osdav.da_transform(x, w, np.zeros(2500))).max() < 1e-11
    with pytest.raises(ValueError):
        dlc.DA([30], 10)


def test_tensor_wrapper_reference_known_answer(dlc, golden):
    """test/TensorflowWrapperTest.py:11-21 run through the MI355X TensorWrapper."""
    tw = dlc.tensor_wrapper
    g = golden("tensorwrapper_test_example.npz")
    x = tw.constant(g["x"])
    w = tw.constant(g["w"])
    y = x.matmul(w.to_tf()).to_tf()
    assert np.array_equal(g["expected"], y.cpu().numpy()) and y.dtype == torch.float64
    # fused matmul.add.sigmoid chain == oracle layer
    from oracle import tensor_ops
    rng = np.random.RandomState(0)
    xb, wb, bb = rng.uniform(0, 1, (3, 30, 40)), rng.standard_normal((40, 25)), rng.standard_normal(25)
    h = tw.constant(xb).corrupt(0).matmul(tw.constant(wb)).add(tw.constant(bb)).sigmoid()
    ref = tensor_ops.sigmoid(tensor_ops.tw_matmul(xb, wb) + bb)
    assert h.shape() == [3, 30, 25] and np.abs(h.numpy() - ref).max() < 1e-12
    assert np.abs(tw.constant(xb).add(tw.constant(xb[0, 0])).sigmoid().numpy() -
                  tensor_ops.sigmoid(xb + xb[0, 0])).max() < 1e-15
    m = tw.random_mask([30, 40], 0.3).numpy()
    assert m.shape == (30, 40) and int((m == 0).sum()) == 360


# --------------------------------------------------------------------------- reference-semantics match
SIM_CASES = ["n6_h8", "n6_h64", "n4_h2500", "n5_h32_params", "n5_p7_h16"]


def sim_dataset(g, name):
    if name + "/dataset" in g:
        return g[name + "/dataset"]
    seed, n, p, h = g[name + "/dataset_seed_uniform01"]
    return np.random.RandomState(int(seed)).uniform(0.0, 1.0, size=(int(n), int(p), int(h)))


@pytest.mark.parametrize("name", SIM_CASES)
def test_similarity_vs_reference_fixture(dlc, golden, name):
    """GPU similarity against the outputs of the reference's own SimilarityCalculator."""
    from oracle import similarity as osim
    g = golden("similarity.npz")
    ds = sim_dataset(g, name)
    mu, sigma, a, b = g[name + "/params"]
    calc = dlc.SimilarityCalculator(ds, mu=mu, sigma=sigma, a=a, b=b)
    want = g[name + "/scores"]
    n = ds.shape[0]
    np.testing.assert_allclose(calc._score.cpu().numpy(), g[name + "/distinctive_score"], rtol=1e-13)
    mf = calc.similarity_matrix(as_int64=False)
    mi = calc.similarity_matrix(as_int64=True)
    assert mi.dtype == np.int64
    for i in range(n):
        assert mf[i, i] == -1 and mi[i, i] == -1
        for j in range(i + 1, n):
            if np.isinf(want[i, j]):
                assert mf[i, j] == want[i, j] and mi[i, j] == osim.INT64_MIN
            else:
                assert abs(mf[i, j] - want[i, j]) <= 1e-9 * abs(want[i, j])
                assert mi[i, j] == int(np.trunc(want[i, j])) or abs(want[i, j] - round(want[i, j])) < 1e-7
            assert mf[j, i] == mf[i, j] and mi[j, i] == mi[i, j]        # mirrored upper triangle
    # the per-pair entry keeps the reference's asymmetry
    for (i, j) in [(0, 1), (1, 0), (2, 3)]:
        got = calc.similarity_score(ds[i], ds[j])
        assert (np.isinf(want[i, j]) and got == want[i, j]) or abs(got - want[i, j]) <= 1e-9 * abs(want[i, j])


def test_similarity_calculator_rereads_its_public_attributes(dlc):
    """The reference reads self.dataset, self.mu, self.sigma on every similarity_score call
    (SimilarityCalculator.py:13-14,25-27); the drop-in hoists the distinctive score, so assigning any of them must
    re-hoist it -- also when the new dataset has another width (the pair buffers follow)."""
    from oracle import similarity as osim
    rng = np.random.RandomState(31)
    ds_a = rng.uniform(0, 1, size=(5, 30, 96))
    ds_b = rng.uniform(0, 1, size=(7, 30, 96)) ** 3
    ds_c = rng.uniform(0, 1, size=(4, 12, 40))
    calc = dlc.SimilarityCalculator(ds_a)
    close = lambda got, want: abs(got - want) <= 1e-9 * abs(want)
    assert close(calc.similarity_score(ds_a[0], ds_a[1]), osim.similarity_score(ds_a, ds_a[0], ds_a[1]))
    calc.dataset = ds_b
    assert calc.dataset is ds_b
    want_b = osim.similarity_score(ds_b, ds_a[0], ds_a[1])
    assert not close(osim.similarity_score(ds_a, ds_a[0], ds_a[1]), want_b)
    assert close(calc.similarity_score(ds_a[0], ds_a[1]), want_b)
    ref = osim.similarity_matrix_f64(ds_b)
    got = calc.similarity_matrix(as_int64=False)
    assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()
    calc.mu, calc.sigma = 0.3, 0.1
    assert close(calc.similarity_score(ds_b[2], ds_b[5]), osim.similarity_score(ds_b, ds_b[2], ds_b[5], mu=0.3, sigma=0.1))
    calc.dataset = ds_c
    assert close(calc.similarity_score(ds_c[0], ds_c[3]), osim.similarity_score(ds_c, ds_c[0], ds_c[3], mu=0.3, sigma=0.1))
    with pytest.raises(ValueError):
        calc.dataset = ds_c[0]


def test_similarity_matrix_vs_oracle_larger(dlc):
    from oracle import similarity as osim
    rng = np.random.RandomState(3)
    ds = rng.uniform(0, 1, size=(23, 30, 300))
    ds[9] = ds[2]
    got = dlc.SimilarityCalculator(ds).similarity_matrix(as_int64=False)
    ref = osim.similarity_matrix_f64(ds)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isposinf(got), np.isposinf(ref))
    assert np.abs(got[fin] - ref[fin]).max() < 1e-9 * np.abs(ref[fin]).max()


def test_similarity_near_ties_follow_the_reference(dlc, monkeypatch):
    """Patches of a frame that differ from each other by 1e-7 in one of their entries: the nearest one is decided far
    below what a Gram-matrix distance |a|^2 + |b|^2 - 2 a.b resolves.  The arg-min filter (csrc/gram_i8.hip) sends
    such candidates to a direct evaluation of |b - a|, which orders them as the reference's np.linalg.norm does."""
    from oracle import similarity as osim
    rng = np.random.RandomState(17)
    for n, p, h in [(6, 30, 64), (9, 30, 250), (5, 32, 2500), (12, 7, 129)]:
        x = 1.0 / (1.0 + np.exp(-35.0 * rng.standard_normal((n * p, h))))
        x[1::2] = x[0::2][: x[1::2].shape[0]]
        x[1::2, 0] += 1e-7
        ds = x.reshape(n, p, h)
        got = dlc.SimilarityCalculator(ds).similarity_matrix(as_int64=False)
        ref = osim.similarity_matrix_f64(ds)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)), (n, p, h)
        assert np.abs(got[fin] - ref[fin]).max() <= 1e-9 * max(1.0, np.abs(ref[fin]).max()), (n, p, h)


def test_similarity_tiny_descriptors_follow_the_reference(dlc, monkeypatch):
    """Two- and three-dimensional saturated descriptors: most patch distances of a frame pair agree to the last bits
    (values 0, 1 and 1 - 1e-12 in every combination), so nearly every arg-min is one the integer products cannot decide
    and many are ties only NumPy's own summation order resolves.  The matrix must still be the reference's."""
    from oracle import similarity as osim
    rng = np.random.RandomState(23)
    for n, p, h in [(12, 6, 3), (9, 30, 2), (14, 5, 1), (7, 32, 9), (10, 17, 8)]:
        ds = 1.0 / (1.0 + np.exp(-35.0 * rng.standard_normal((n, p, h))))
        got = dlc.SimilarityCalculator(ds).similarity_matrix(as_int64=False)
        ref = osim.similarity_matrix_f64(ds)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isnan(got), np.isnan(ref)), (n, p, h)
        assert np.abs(got[fin] - ref[fin]).max() <= 1e-9 * max(1.0, np.abs(ref[fin]).max()), (n, p, h)
    # zeros and ones with 1e-13 of noise: every squared distance an integer up to the last bits, ties among the
    # candidates of nearly every arg-min, at widths of one, several and many leaves of the pairwise summation
    for n, p, h, noise in [(10, 25, 78, 1e-13), (6, 20, 300, 1e-13), (4, 30, 2500, 1e-13), (5, 11, 129, 1e-13), (5, 8, 1031, 1e-13),
                           (8, 30, 64, 0.0), (5, 20, 300, 0.0), (6, 32, 17, 0.0)]:     # noise 0: purely binary (integer sums: exact in any order)
        ds = (rng.uniform(size=(n, p, h)) < 0.5).astype(np.float64) + noise * rng.uniform(size=(n, p, h))
        if noise == 0.0:
            ds *= rng.randint(1, 4)                       # ... and small integers
        got = dlc.SimilarityCalculator(ds).similarity_matrix(as_int64=False)
        ref = osim.similarity_matrix_f64(ds)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)), (n, p, h)
        assert np.abs(got[fin] - ref[fin]).max() <= 1e-9 * max(1.0, np.abs(ref[fin]).max()), (n, p, h)


def test_similarity_filter_equals_fp64_gram_route(eng, monkeypatch):
    """The two routes of dlc_sdav_similarity_matrix -- exact integer products of column-centred 24-bit fixed-point descriptors that
    decide the arg-min (with a direct fp64 evaluation where their error bound cannot), and the fp64 Gram matrix --
    give the same matrix bit for bit: uniform, normal (negative values, range far from [0, 1]), saturated sigmoid
    outputs, duplicated patches (exact ties: first index), constant data, ragged sizes."""
    g = torch.Generator(device=eng.device); g.manual_seed(5)
    rng = np.random.RandomState(5)

    def both(ds):
        score = eng.distinctive_score(ds, 0.5, 0.2)
        out = []
        for force in (True, False):                    # DLC_SIM_FORCE_F64: the fp64 Gram form; then the arg-min filter
            mf, mi = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=force)
            out.append((mf.clone(), mi.clone()))
        return out

    # (H a multiple of 256: the LAST k-step of the product kernel is all data, not zero padding -- what an accumulator read
    # too early behind the last MFMAs would lose, docs/LAB.md 9.2)
    for n, p, h in [(2, 1, 8), (3, 7, 64), (6, 30, 8), (20, 30, 250), (40, 32, 2500), (70, 13, 129), (300, 30, 256), (150, 5, 1000),
                    (40, 30, 768), (40, 31, 256), (40, 16, 1024)]:
        sat = torch.sigmoid(35.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64))
        dup = sat.clone().reshape(n * p, h)
        if n * p > 4:
            src = torch.from_numpy(rng.randint(0, n * p, size=n * p // 3 + 1)).to(eng.device)
            dst = torch.from_numpy(rng.randint(0, n * p, size=n * p // 3 + 1)).to(eng.device)
            dup[dst] = dup[src].clone()
        twins = sat.clone()                           # copies INSIDE every frame: exact ties in a fifth of the arg-mins,
        if p >= 6:                                    # resolved by the rows' content hashes without reading a row
            twins[:, 1] = twins[:, 0]; twins[:, 3] = twins[:, 2]; twins[:, 5] = twins[:, 4]
        for name, ds in (("uniform", torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)),
                         ("normal", 3.0 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64) - 1.0),
                         ("saturated", sat), ("duplicates", dup.reshape(n, p, h)), ("twins", twins)):
            (f_ref, i_ref), (f_got, i_got) = both(ds)
            assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref), (name, n, p, h)
    (f_ref, i_ref), (f_got, i_got) = both(torch.full((5, 30, 64), 0.25, device=eng.device, dtype=torch.float64))
    assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref)
    bad = torch.rand((6, 30, 64), generator=g, device=eng.device, dtype=torch.float64)
    bad[2, 3, 5] = float("nan")                       # a NaN in the data: the filter hands the call to the fp64 route
    (f_ref, i_ref), (f_got, i_got) = both(bad)
    assert torch.equal(i_got, i_ref) and torch.equal(f_got.isnan(), f_ref.isnan())
    assert torch.equal(torch.nan_to_num(f_got), torch.nan_to_num(f_ref))
    # ... and under DLC_SIM_NO_HOST_SYNC (no flag read on the host, graph-capturable) the call SAYS that it could not:
    # NaN / INT64_MIN everywhere and stats[1] = 1; on finite data it is the same matrix, stats[0] = the direct evaluations
    stats = torch.full((2,), -7, dtype=torch.int64, device=eng.device)
    score = eng.distinctive_score(bad, 0.5, 0.2)
    f_ns, i_ns = eng.sdav_similarity_matrix(bad, score, 10.0, -10.0, no_host_sync=True, stats=stats)
    assert bool(f_ns.isnan().all()) and bool((i_ns == torch.iinfo(torch.int64).min).all()) and stats.tolist()[1] == 1
    good = torch.rand((6, 30, 64), generator=g, device=eng.device, dtype=torch.float64)
    score = eng.distinctive_score(good, 0.5, 0.2)
    f_a, i_a = (t.clone() for t in eng.sdav_similarity_matrix(good, score, 10.0, -10.0))
    f_b, i_b = eng.sdav_similarity_matrix(good, score, 10.0, -10.0, no_host_sync=True, stats=stats)
    assert torch.equal(f_a, f_b) and torch.equal(i_a, i_b) and stats.tolist()[1] == 0 and stats.tolist()[0] >= 0


def _low_contrast(shape, g, device, spread=1e-3):
    """Descriptors whose columns each stay within ~spread of their own mean while the means spread over [0.15, 0.88]: what
    real frames give through 1/sqrt(fan_in) weights."""
    n, p, h = shape
    means = 0.15 + 0.73 * torch.rand((h,), generator=g, device=device, dtype=torch.float64)
    return means + spread * torch.randn(shape, generator=g, device=device, dtype=torch.float64)


def _unkey(k):
    """dlc_f64_key inverted (csrc/dlc_internal.h): ordered 64-bit keys -> doubles."""
    k = k.astype(np.uint64)
    top = (k >> np.uint64(63)).astype(bool)
    u = np.where(top, k & np.uint64(0x7fffffffffffffff), ~k)
    return u.view(np.float64)


@pytest.mark.parametrize("rows,h", [(1000, 40), (193, 18), (192, 16), (5, 7), (391, 33), (2 * 192 + 1, 2500), (600, 2)])
def test_distinctive_score_pass_is_the_row_ordered_chain(eng, rows, h):
    """dlc_sdav_distinctive_score: the column mean is NumPy's row-by-row accumulation (SimilarityCalculator.py:20-23) --
    made visible with columns of cancelling values (1e16, 1, -1e16, 3, ...: any other summation order changes the mean by
    whole units) -- on both forms of the pass (the LDS-DMA one: even H; the register-staged one: odd H), at batch edges
    (192 rows a batch), column counts that end inside a 16-column group; the columns' extremes and the NaN flag it leaves
    for the similarity's filter are exact."""
    from oracle import similarity as osim
    rng = np.random.RandomState(rows * 7 + h)
    x = rng.uniform(0.0, 1.0, size=(rows, h))
    for c in range(h):                                                 # as many + 1e16 as - 1e16 in every column, anywhere
        pos = rng.permutation(rows)[:2 * (rows // 10)]
        x[pos[0::2], c], x[pos[1::2], c] = 1e16, -1e16
    want_avg = np.cumsum(x, axis=0)[-1] / rows                          # cumsum IS the sequential chain
    assert np.array_equal(want_avg, osim.average_response(x.reshape(rows, 1, h)))
    mu, sigma = float(np.median(want_avg)), 0.3
    want = np.exp(-((want_avg - mu) ** 2) / (2.0 * sigma * sigma))
    score, r = eng.distinctive_score(torch.from_numpy(x).to(eng.device), mu, sigma, with_range=True)
    np.testing.assert_allclose(score.cpu().numpy(), want, rtol=1e-12)
    # a different order would show: the pairwise sum of the same column differs from the chain in most columns
    assert (np.cumsum(x[::-1], axis=0)[-1] / rows != want_avg).mean() > 0.3 or rows < 16
    words = r.cpu().numpy().view(np.uint64)
    assert words[2] == 0
    assert np.array_equal(_unkey(words[3:3 + h]), x.min(axis=0)) and np.array_equal(_unkey(words[3 + h:3 + 2 * h]), x.max(axis=0))
    x[rows // 2, h - 1] = np.nan
    _, r = eng.distinctive_score(torch.from_numpy(x).to(eng.device), mu, sigma, with_range=True)
    assert r.cpu().numpy().view(np.uint64)[2] == 1


def test_similarity_filter_low_contrast_columns(eng):
    """The filter quantises x - centre of the COLUMN (a per-column offset changes no distance): low-contrast descriptors,
    on which a global range left every arg-min inside the error window, are decided by the int8 products --
    filter == fp64 form bit for bit, the filter keeps the call, and only a small fraction of arg-mins is evaluated directly;
    also with a distinctive-score range handed over, and for columns that are constant (range 0) next to live ones."""
    g = torch.Generator(device=eng.device); g.manual_seed(31)
    for n, p, h in [(40, 30, 2500), (60, 13, 300), (30, 32, 1024)]:
        ds = _low_contrast((n, p, h), g, eng.device)
        ds[:, :, 5] = 0.321                                               # a constant column
        score, rng = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)
        stats = torch.zeros((2,), dtype=torch.int64, device=eng.device)
        f_ref, i_ref = (t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=True))
        for r in (None, rng):
            f_got, i_got = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, stats=stats, range=r)
            assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref), (n, p, h)
            direct, why = stats.tolist()
            assert why == 0 and direct <= 0.02 * (n * (n - 1) // 2 * p), (n, p, h, direct, why)


def test_similarity_filter_exit_to_fp64_route(eng):
    """The filter's exit: one column with a range a million times the others' leaves every distance inside the error window
    (the scale is the LARGEST column range).  The sample taken before the product kernel says so, the call takes the fp64
    Gram form -- same matrix as DLC_SIM_FORCE_F64, == the oracle -- and reports it (stats = [0, 2]); under
    DLC_SIM_NO_HOST_SYNC nobody can act on a verdict, the filter runs and evaluates its arg-mins directly: the same matrix."""
    from oracle import similarity as osim
    g = torch.Generator(device=eng.device); g.manual_seed(32)
    n, p, h = 24, 30, 512
    ds = 0.5 + 1e-7 * torch.randn((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
    ds[:, :, 0] = torch.rand((n, p), generator=g, device=eng.device, dtype=torch.float64) < 0.001   # a rare spike column: 0 / 1
    ds[0, 0, 0] = 1.0
    score = eng.distinctive_score(ds, 0.5, 0.2)
    stats = torch.full((2,), -7, dtype=torch.int64, device=eng.device)
    f_ref, i_ref = (t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=True))
    f_got, i_got = (t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, stats=stats))
    assert stats.tolist() == [0, 2]
    assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref)
    f_ns, i_ns = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, stats=stats, no_host_sync=True)
    direct, why = stats.tolist()
    assert why == 0 and direct > 0.5 * (n * (n - 1) // 2 * p)
    assert torch.equal(f_ns, f_ref) and torch.equal(i_ns, i_ref)
    dsn, sc, mf = ds.cpu().numpy(), score.cpu().numpy(), f_got.cpu().numpy()
    rng = np.random.RandomState(3)
    for _ in range(12):
        i, j = sorted(rng.choice(n, 2, replace=False))
        d = osim.weighted_distances(dsn[i], dsn[j], osim.match_features(dsn[i], dsn[j]), sc)
        want = np.sum(10 - 10 * np.log(d))
        assert abs(mf[i, j] - want) <= 1e-9 * abs(want), (i, j)
    # benign data: the sample lets the filter keep the call
    ok = torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
    eng.sdav_similarity_matrix(ok, eng.distinctive_score(ok, 0.5, 0.2), 10.0, -10.0, stats=stats)
    assert stats.tolist()[1] == 0


def test_similarity_filter_in_row_chunks(eng, monkeypatch):
    """The product block in several row chunks (chunk_bytes shrinks the 8 GiB cap): chunk origins that are not
    multiples of the panels' 16-patch groups, for patch counts that are and are not; the same matrix as in one chunk."""
    g = torch.Generator(device=eng.device); g.manual_seed(11)
    for n, p, h in [(150, 30, 64), (90, 7, 130), (64, 32, 256), (200, 13, 40)]:
        ds = torch.rand((n, p, h), generator=g, device=eng.device, dtype=torch.float64)
        score = eng.distinctive_score(ds, 0.5, 0.2)
        f_one, i_one = (t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0))
        for frames_per_chunk in (1, 3, 17):
            cb = max(1 << 16, frames_per_chunk * p * (n * p + 20) * 4 + 64)
            f_c, i_c = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, chunk_bytes=cb)
            assert torch.equal(f_c, f_one) and torch.equal(i_c, i_one), (n, p, h, frames_per_chunk)


def test_similarity_filter_accumulator_extremes(eng, monkeypatch):
    """The widest descriptors the filter takes (H = 32768) with rows of ones against rows of ones: every slice of every
    element 127, the class-4 accumulators at 3 * 32768 * 127^2 = 1.59e9 of the int32 range; one frame, two frames, a
    single patch per frame; results equal to the fp64 Gram form."""
    g = torch.Generator(device=eng.device); g.manual_seed(2)

    def both(ds):
        score = eng.distinctive_score(ds, 0.5, 0.2)
        out = []
        for force in (True, False):                    # DLC_SIM_FORCE_F64: the fp64 Gram form; then the arg-min filter
            mf, mi = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, force_f64=force)
            out.append((mf.clone(), mi.clone()))
        return out

    h = 32768
    ds = (torch.rand((5, 3, h), generator=g, device=eng.device, dtype=torch.float64) < 0.5).double()
    ds[0, 0] = 1.0; ds[1, 1] = 1.0; ds[2, 0] = 1.0; ds[3, 2] = 0.0; ds[4, 1] = 1.0
    ds += 1e-3 * torch.rand((5, 3, h), generator=g, device=eng.device, dtype=torch.float64)
    ds.clamp_(0.0, 1.0)
    (f_ref, i_ref), (f_got, i_got) = both(ds)
    assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref)
    for n, p, hh in [(1, 30, 64), (2, 30, 64), (2, 1, 1), (3, 32, 33), (40, 1, 700)]:
        ds = torch.rand((n, p, hh), generator=g, device=eng.device, dtype=torch.float64)
        (f_ref, i_ref), (f_got, i_got) = both(ds)
        assert torch.equal(f_got, f_ref) and torch.equal(i_got, i_ref), (n, p, hh)


def test_distance_vs_reference_fixture(dlc, golden):
    g = golden("distance.npz")
    dc = dlc.DistanceCalculator
    assert dc.calculate_distance(g["probe/a"], g["probe/b"]) == 6
    per = [dc.calculate_distance([v], [np.int8(0)]) for v in g["all/a"][::17]]
    assert per == g["all/per_element"][::17].tolist()
    for name in ("n7_d2243", "n9_d37", "n3_d1"):
        m = dc.distance_matrix(g[name + "/desc"])
        assert m.dtype == np.int64 and np.array_equal(m, g[name + "/matrix"])


def test_distance_matrix_vs_oracle_larger(dlc):
    from oracle import distance as odist
    rng = np.random.RandomState(13)
    desc = rng.randint(-128, 128, size=(203, 2243)).astype(np.int8)
    desc[50] = desc[3]
    got = dlc.DistanceCalculator.distance_matrix(desc)
    assert np.array_equal(got, odist.distance_matrix(desc))
    assert got[50, 3] == 0 and np.array_equal(got, got.T)


def test_encoders_fuzz(dlc, eng):
    """Seeded odd sizes: dlc_sdav_encode with 1-5 layers of arbitrary widths, CnnVtl at non-reference
    frame sizes (down to the smallest that survives both pools)."""
    from oracle import sdav as osdav, cnn_vtl as ocnn
    rng = np.random.RandomState(5)
    for _ in range(8):
        layers = int(rng.randint(1, 6))
        dims = [int(rng.randint(1, 260)) for _ in range(layers + 1)]
        rows = int(rng.randint(1, 400))
        ws = [rng.standard_normal((dims[l], dims[l + 1])) for l in range(layers)]
        bs = [rng.standard_normal(dims[l + 1]) for l in range(layers)]
        x = rng.uniform(0, 1, (rows, 1, dims[0]))
        got = eng.sdav_encode(torch.from_numpy(x.reshape(rows, dims[0])).to(eng.device),
                              [torch.from_numpy(w).to(eng.device) for w in ws], [torch.from_numpy(b).to(eng.device) for b in bs])
        assert np.abs(got.cpu().numpy() - osdav.transform(x, ws, bs)).max() < 1e-10, dims
    for h, w, n in [(35, 39, 2), (67, 83, 3), (50, 131, 1)]:
        frames = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.float64)
        net = dlc.CnnVtl(input_shape=[n, h, w, 3], seed=11, mask_seed=12, compress_factor=90.0)
        cols = ocnn.column_indices(ocnn.layer_sizes((h, w)), 90.0, seed=12)
        assert np.array_equal(net.columns, cols)
        cw, cb = ocnn.init_weights(11)
        got, ref = net.transform(frames), ocnn.transform(frames, cw, cb, cols)
        diff = (got.astype(np.int16) - ref.astype(np.int16)) % 256
        assert got.shape == ref.shape and np.count_nonzero(diff) <= 2 and np.all((diff == 0) | (diff == 1) | (diff == 255)), (h, w)


def test_match_reference_semantics_fuzz(dlc):
    """Seeded random sizes for the two reference-semantics matrices: ragged N / D / P / H, 1-frame and
    1-byte cases, duplicates (distance 0, similarity +inf), int8 extremes."""
    from oracle import distance as odist, similarity as osim
    rng = np.random.RandomState(99)
    for n, d in [(1, 1), (2, 3), (65, 64), (64, 65), (129, 17), (33, 4099), (7, 1), (200, 255)]:
        desc = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        if n > 2:
            desc[n - 1] = desc[0]
            desc[1] = -128
        got = dlc.DistanceCalculator.distance_matrix(desc)
        assert got.dtype == np.int64 and np.array_equal(got, odist.distance_matrix(desc)), (n, d)
    for n, p, h in [(2, 1, 1), (3, 30, 7), (9, 64, 33), (17, 5, 129), (4, 30, 2500), (31, 2, 64)]:
        ds = rng.uniform(0, 1, size=(n, p, h))
        if n > 2:
            ds[n - 1] = ds[1]
        got = dlc.SimilarityCalculator(ds).similarity_matrix(as_int64=False)
        ref = osim.similarity_matrix_f64(ds)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isposinf(got), np.isposinf(ref)) and np.array_equal(np.isnan(got), np.isnan(ref)), (n, p, h)
        assert np.abs(got[fin] - ref[fin]).max() <= 1e-9 * max(1.0, np.abs(ref[fin]).max()), (n, p, h)
        assert np.array_equal(dlc.SimilarityCalculator(ds).similarity_matrix(), osim.similarity_matrix(ds)), (n, p, h)


# --------------------------------------------------------------------------- CnnVtl
def test_cnn_vtl_pieces_vs_oracle(eng):
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(0)
    x = rng.uniform(0, 255, size=(2, 23, 31, 5))
    xt = torch.from_numpy(x).to(eng.device)
    assert np.array_equal(eng.maxpool3x3s2(xt).cpu().numpy(), ocnn.maxpool3x3s2(x))
    w = rng.standard_normal((3, 3, 5, 7))
    b = rng.standard_normal(7)
    for stride, pad in ((1, "SAME"), (2, "VALID"), (2, "SAME")):
        oh, ph = ocnn._out_size(23, 3, stride, pad)
        ow, pw = ocnn._out_size(31, 3, stride, pad)
        cols = eng.im2col(xt, 3, 3, stride, ph, pw, oh, ow)
        y = eng.gemm_bias_act(cols, torch.from_numpy(w.reshape(45, 7)).to(eng.device), torch.from_numpy(b).to(eng.device),
                              act=2).reshape(2, oh, ow, 7).cpu().numpy()
        ref = ocnn.conv2d_nhwc(x, w, b, stride, pad, True)
        assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-10


def test_conv2d_implicit_gemm_vs_oracle(eng):
    """dlc_conv2d_nhwc_f64 (no im2col matrix) for the cnn_vtl layer shapes with C % 8 == 0."""
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(4)
    for (h, w, c, kh, cout, stride, pad, relu) in ((22, 28, 96, 5, 256, 1, "SAME", True), (10, 13, 256, 3, 384, 1, "SAME", True),
                                                   (10, 13, 384, 3, 256, 1, "SAME", False), (17, 19, 8, 3, 5, 2, "VALID", True),
                                                   (9, 9, 16, 5, 130, 2, "SAME", False)):
        x = rng.standard_normal((3, h, w, c))
        wk = rng.standard_normal((kh, kh, c, cout)) / np.sqrt(kh * kh * c)
        b = rng.standard_normal(cout)
        oh, ph = ocnn._out_size(h, kh, stride, pad)
        ow, pw = ocnn._out_size(w, kh, stride, pad)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
        got = eng.conv2d(dev(x), dev(wk.reshape(-1, cout)), dev(b), kh, kh, stride, ph, pw, oh, ow, 2 if relu else 0)
        ref = ocnn.conv2d_nhwc(x, wk, b, stride, pad, relu)
        assert tuple(got.shape) == ref.shape and np.abs(got.cpu().numpy() - ref).max() < 1e-10
        cols = eng.im2col(dev(x), kh, kh, stride, ph, pw, oh, ow)                  # explicit path: bit-identical sums
        alt = eng.gemm_bias_act(cols, dev(wk.reshape(-1, cout)), dev(b), act=2 if relu else 0).reshape(got.shape)
        assert torch.equal(alt, got)
    # channel counts that are not multiples of 8 (conv1: 3 input channels) take the element-wise loader
    for (h, w, c, kh, cout, stride, pad, relu) in ((27, 31, 3, 11, 96, 4, "VALID", True), (9, 11, 5, 3, 7, 1, "SAME", False),
                                                   (8, 8, 1, 1, 3, 2, "SAME", True), (12, 10, 12, 5, 130, 2, "SAME", True)):
        x = rng.standard_normal((2, h, w, c))
        wk = rng.standard_normal((kh, kh, c, cout)) / np.sqrt(kh * kh * c)
        b = rng.standard_normal(cout)
        oh, ph = ocnn._out_size(h, kh, stride, pad)
        ow, pw = ocnn._out_size(w, kh, stride, pad)
        got = eng.conv2d(dev(x), dev(wk.reshape(-1, cout)), dev(b), kh, kh, stride, ph, pw, oh, ow, 2 if relu else 0)
        assert np.abs(got.cpu().numpy() - ocnn.conv2d_nhwc(x, wk, b, stride, pad, relu)).max() < 1e-10, (h, w, c, kh)
        cols = eng.im2col(dev(x), kh, kh, stride, ph, pw, oh, ow)
        alt = eng.gemm_bias_act(cols, dev(wk.reshape(-1, cout)), dev(b), act=2 if relu else 0).reshape(got.shape)
        assert torch.equal(alt, got)


def test_conv2d_dma_kernel_padding_and_stride_fuzz(eng):
    """Convolutions large enough for the LDS-DMA kernel (C % 16 == 0, >= 16 tiles) whose padding is served by
    out-of-range buffer offsets: SAME / VALID, strides 1-3, kernels 1-7, odd image sizes, tiles that span several
    images -- against the oracle and, bit for bit, the explicit im2col + GEMM path."""
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(9)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    for (n, h, w, c, kh, cout, stride, pad) in ((40, 17, 19, 16, 3, 40, 2, "VALID"), (70, 12, 10, 32, 5, 130, 2, "SAME"),
                                                (90, 9, 9, 16, 5, 96, 1, "SAME"), (25, 21, 23, 48, 7, 64, 3, "SAME"),
                                                (33, 14, 15, 64, 1, 256, 1, "VALID"), (300, 5, 6, 16, 3, 34, 1, "SAME"),
                                                (16, 31, 29, 16, 3, 8, 1, "SAME")):
        x = rng.standard_normal((n, h, w, c))
        wk = rng.standard_normal((kh, kh, c, cout)) / np.sqrt(kh * kh * c)
        b = rng.standard_normal(cout)
        oh, ph = ocnn._out_size(h, kh, stride, pad)
        ow, pw = ocnn._out_size(w, kh, stride, pad)
        got = eng.conv2d(dev(x), dev(wk.reshape(-1, cout)), dev(b), kh, kh, stride, ph, pw, oh, ow, 2)
        ref = ocnn.conv2d_nhwc(x[:4], wk, b, stride, pad, True)
        assert np.abs(got[:4].cpu().numpy() - ref).max() < 1e-10, (n, h, w, c, kh)
        cols = eng.im2col(dev(x), kh, kh, stride, ph, pw, oh, ow)
        alt = eng.gemm_bias_act(cols, dev(wk.reshape(-1, cout)), dev(b), act=2).reshape(got.shape)
        assert torch.equal(alt, got), (n, h, w, c, kh)


def test_conv2d_frame_minmax_keys(eng):
    """dlc_conv2d_nhwc_f64_stats: the per-frame minimum / maximum folded while the convolution runs == min / max of its
    output, exactly, on every route: the LDS-DMA kernel's epilogue (large launches, signed values, a tile spanning two
    frames), the pass over the output (small launches, fewer than 64 output pixels per frame), several layers into one
    set of keys; and quant_gather from those keys == the one-call minmax_quant_gather, byte for byte."""
    g = torch.Generator(device=eng.device); g.manual_seed(12)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=eng.device, dtype=torch.float64)
    for (n, h, w, c, k, cout, pad, act) in ((40, 22, 28, 96, 5, 256, 2, 0),      # 40 * 616 rows: LDS-DMA kernel, no ReLU
                                            (300, 10, 13, 256, 3, 384, 1, 2),     # 130 pixels per frame: two frames per wave tile
                                            (3, 10, 13, 32, 3, 24, 1, 0),         # small launch: register-staged kernel + pass
                                            (700, 6, 7, 32, 3, 128, 1, 0)):       # 42 pixels per frame: more than two per wave
        x, wk, b = rnd(n, h, w, c), rnd(k * k * c, cout) / (k * np.sqrt(c)), rnd(cout)
        keys = eng.frame_minmax_keys(n)
        y = eng.conv2d(x, wk, b, k, k, 1, pad, pad, h, w, act, frame_keys=keys)
        assert torch.equal(y, eng.conv2d(x, wk, b, k, k, 1, pad, pad, h, w, act))
        cols = torch.arange(0, y[0].numel(), 7, device=eng.device)
        got = eng.quant_gather([y], cols, keys)
        assert torch.equal(got, eng.minmax_quant_gather([y], cols)), (n, h, w, c)
        flat = y.reshape(n, -1).cpu().numpy()                   # NumPy: torch's scalar / tensor is reciprocal * scalar
        mn, mx = flat.min(axis=1, keepdims=True), flat.max(axis=1, keepdims=True)
        scaled = (flat[:, cols.cpu().numpy()] - mn) * (255.0 / (mx - mn))
        assert np.array_equal(got.cpu().numpy(), np.trunc(scaled).astype(np.int64).astype(np.int8))
    # two layers into one set of keys
    x, w1, b1 = rnd(64, 12, 16, 16), rnd(9 * 16, 64) / 12, rnd(64)
    w2, b2 = rnd(9 * 64, 32) / 24, rnd(32)
    keys = eng.frame_minmax_keys(64)
    y1 = eng.conv2d(x, w1, b1, 3, 3, 1, 1, 1, 12, 16, 2, frame_keys=keys)
    y2 = eng.conv2d(y1, w2, b2, 3, 3, 1, 1, 1, 12, 16, 0, frame_keys=keys)
    cols = torch.arange(0, y1[0].numel() + y2[0].numel(), 11, device=eng.device)
    assert torch.equal(eng.quant_gather([y1, y2], cols, keys), eng.minmax_quant_gather([y1, y2], cols))
    with pytest.raises(ValueError):
        eng.conv2d(x, w1, b1, 3, 3, 1, 1, 1, 12, 16, 2, frame_keys=keys[:10])


def test_cnn_vtl_transform_vs_oracle(dlc):
    """CnnVtl.transform at the reference's 192x240 frame size, seeded weights + mask."""
    from oracle import cnn_vtl as ocnn
    rng = np.random.RandomState(2)
    x = rng.randint(0, 256, size=(3, 192, 240, 3)).astype(np.float64)      # uint8 pixels fed as fp64
    net = dlc.CnnVtl(input_shape=[3, 192, 240, 3], seed=5, mask_seed=9)
    assert net.layer_sizes == [256128, 157696, 49920, 49920, 33280] == ocnn.layer_sizes((192, 240))
    cols = ocnn.column_indices(net.layer_sizes, 99.59, seed=9)
    assert np.array_equal(net.columns, cols) and cols.size <= 2243
    ws, bs = ocnn.init_weights(5)
    got = net.transform(x)
    ref = ocnn.transform(x, ws, bs, cols)
    assert got.dtype == np.int8 and got.shape == ref.shape == (3, cols.size)
    # fp64 on both sides, different summation order: a value within ~1e-11 of an integer could
    # truncate differently.  None does for this seed: every one of the ~6700 bytes is identical.
    assert np.array_equal(got, ref)
    assert (ref < 0).any() and (ref > 0).any()                       # the wrap (>127 -> negative) is exercised


def test_cnn_vtl_load_alexnet_npy_grouped_layout(dlc, tmp_path):
    """load_alexnet_npy on a synthetic blob in bvlc_alexnet.npy's REAL layout (cnn_vtl.py:137-149): a pickled
    {layer: [W, b]} dict with fc6-8 present and AlexNet's grouped kernels for conv2 / conv4 / conv5, which the
    reference pushes through tf.constant_initializer into ungrouped variables (values in C order, the last one
    repeated).  The loaded network == the oracle on the oracle's own fill of the same dict, byte for byte."""
    from oracle import cnn_vtl as ocnn
    from test_host_logic_cpu import _grouped_alexnet_dict
    rng = np.random.RandomState(17)
    d = _grouped_alexnet_dict(rng)
    path = str(tmp_path / "bvlc_alexnet.npy")
    np.save(path, d, allow_pickle=True)
    net = dlc.CnnVtl(input_shape=[2, 192, 240, 3], seed=1, mask_seed=6)
    before = net.transform(np.ones((1, 192, 240, 3)))
    net.load_alexnet_npy(path)
    x = rng.randint(0, 256, size=(2, 192, 240, 3)).astype(np.float64)
    ws, bs = ocnn.weights_from_alexnet_dict(d)
    assert ws[1].shape == (5, 5, 96, 256) and ws[3].shape == (3, 3, 384, 384) and ws[4].shape == (3, 3, 384, 256)
    got = net.transform(x)
    ref = ocnn.transform(x, ws, bs, ocnn.column_indices(net.layer_sizes, 99.59, seed=6))
    assert np.array_equal(got, ref)
    assert not np.array_equal(net.transform(np.ones((1, 192, 240, 3))), before)      # the weights really changed
    with pytest.raises(ValueError):                                                    # more values than the variable holds
        bad = dict(d)
        bad["conv1"] = [np.zeros((11, 11, 4, 96), dtype=np.float32), d["conv1"][1]]
        np.save(path, bad, allow_pickle=True)
        net.load_alexnet_npy(path)


def test_cnn_vtl_surface(dlc):
    net = dlc.CnnVtl()                                               # reference default 224x224
    assert net.input_shape == [1, 224, 224, 3] and net.compress_factor == 99.59
    assert net.layer_sizes == [279936, 173056, 55296, 55296, 36864]
    y1 = net.transform(np.ones((1, 224, 224, 3)))                    # basic_example.py:6-13
    y2 = y1.copy()
    y2[0, 0] = 23
    d = dlc.DistanceCalculator.calculate_distance(y1[0], y2[0])
    assert y1.shape == (1, net.columns.size) and d >= 0
    with pytest.raises(ValueError):
        net.transform(np.ones((1, 100, 100, 3)))


def test_pipeline_and_coop_select_equal_one_shot(eng, dlc):
    """Two-stream MatchPipeline (cooperative select kernel) == the one-shot call, batch after batch."""
    rng = np.random.RandomState(21)
    n, d, nq, k = 70000, 512, 200, 20
    db = dlc.KeyframeDatabase(rng.standard_normal((n, d)).astype(np.float32), dtype="bf16", row_offset=5)
    batches = [db.prepare_queries(rng.standard_normal((nq, d)).astype(np.float32)) for _ in range(5)]
    want = [tuple(t.clone() for t in db.match_topk(q, k)) for q in batches]
    pipe = dlc.MatchPipeline(db, k, depth=2)
    tickets, got = [], []
    for j, q in enumerate(batches):
        tickets.append(pipe.submit(q))
        if j >= 1:                                   # fetch with one batch of lag, as a server would
            got.append(tuple(t.clone() for t in pipe.result(tickets[j - 1])))
    got.append(tuple(t.clone() for t in pipe.result(tickets[-1])))
    for (ws, wi), (gs, gi) in zip(want, got):
        assert torch.equal(wi, gi) and torch.equal(ws, gs)
    # stage-wise C ABI with the cooperative kernel on ONE stream as well
    ws_buf = torch.empty(eng.topk_workspace_bytes(nq, n, d, k), dtype=torch.uint8, device=eng.device)
    s = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
    i = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
    eng.score_groups(batches[0], db.rows, k, ws_buf)
    eng.select_topk(batches[0], db.rows, k, ws_buf, s, i, row_offset=5, coop=True)
    assert torch.equal(i, want[0][1]) and torch.equal(s, want[0][0])


@pytest.mark.parametrize("n", [50000, 80000])
def test_group_exchange_protocol_equals_unsharded(eng, dlc, n):
    """The sharded protocol of MatchPipeline, emulated with 4 shards on one GPU: select groups per
    shard, 'all-gather' their maxima, filtered re-score per shard, packed merge == one shard.
    (50 000: 12 500-row shards, whose one-shot call takes the small-database plan; 80 000: 20 000-row shards.)"""
    rng = np.random.RandomState(33)
    d, nq, k, parts = 256, 96, 20, 4
    x = rng.standard_normal((n, d)).astype(np.float32)
    x[[100, 20000, 20001, 40000]] = x[7]                       # exact ties across shards
    db = stored(eng, x, "bf16")
    q = stored(eng, np.concatenate([x[[7, 9]], rng.standard_normal((nq - 2, d)).astype(np.float32)]), "bf16")
    want = eng.match_topk(q, db, k, details=True)
    kg = eng.groups_per_query(k)
    ids, mx, shards, wss = [], [], [], []
    for r in range(parts):
        lo, hi = dlc.shard_bounds(n, parts, r)
        ws = torch.empty(eng.topk_workspace_bytes(nq, hi - lo, d, k), dtype=torch.uint8, device=eng.device)
        gi = torch.empty((nq, kg), dtype=torch.int32, device=eng.device)
        gm = torch.empty((nq, kg + 1), dtype=torch.float32, device=eng.device)
        eng.score_groups(q, db[lo:hi], k, ws)
        eng.select_groups(q, db[lo:hi], k, ws, gi, gm, coop=(r % 2 == 1))
        ids.append(gi), mx.append(gm), shards.append((lo, hi)), wss.append(ws)
    all_max = torch.stack(mx)                                  # [parts, nq, kg + 1]; column kg = what a shard leaves behind
    lists = all_max[:, :, :kg]
    gathered = torch.empty((parts, nq * k * 16), dtype=torch.uint8, device=eng.device)
    bounds = []
    kept = 0
    for r, (lo, hi) in enumerate(shards):
        idx = gathered[r, :nq * k * 8].view(torch.int64).view(nq, k)
        sc = gathered[r, nq * k * 8:].view(torch.float64).view(nq, k)
        bound = torch.empty((nq,), dtype=torch.float32, device=eng.device)
        eng.rescore_topk(q, db[lo:hi], k, ids[r], mx[r], sc, idx, bound=bound, all_max=all_max, row_offset=lo, coop=(r % 2 == 0))
        bounds.append(bound)
        # no returned row may come from a group the filter must drop, and the filter must bite
        grp = torch.div(idx - lo, 8, rounding_mode="floor")                                 # [nq, k]
        pos = (ids[r].long().unsqueeze(1) == grp.unsqueeze(2))                              # [nq, k, kg]
        valid = idx >= 0                                                                    # empty slots: too few survivors
        assert bool((pos.any(dim=2) | ~valid).all())
        gval = (mx[r][:, :kg].unsqueeze(1) * pos).sum(dim=2)                                # that group's maximum
        greater = (lists.permute(1, 0, 2).reshape(nq, 1, -1) > gval.unsqueeze(2)).sum(dim=2)
        assert bool(((greater < kg) | ~valid).all())
        own = (lists.permute(1, 0, 2).reshape(nq, 1, -1) > mx[r][:, :kg].unsqueeze(2)).sum(dim=2) < kg   # [nq, kg]
        kept += int(own.sum())
        # unfiltered re-score of the same list == the fused one-shot result of that shard, bit for bit
        s2 = torch.empty((nq, k), dtype=torch.float64, device=eng.device)
        i2 = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
        eng.rescore_topk(q, db[lo:hi], k, ids[r], mx[r], s2, i2, all_max=None, row_offset=lo)
        one = eng.match_topk(q, db[lo:hi], k, row_offset=lo, details=True)
        assert torch.equal(one.idx, i2) and torch.equal(one.scores_f64, s2)
    # every shard computes the same bound: the best maximum left behind anywhere (dropped groups + the shards' rests)
    assert all(torch.equal(b, bounds[0]) for b in bounds)
    flat = lists.permute(1, 0, 2).reshape(nq, -1)
    cnt = (flat.unsqueeze(1) > flat.unsqueeze(2)).sum(dim=2)                                # strictly larger maxima per entry
    dropped = torch.where(cnt >= kg, flat, torch.full_like(flat, float("-inf"))).max(dim=1).values
    assert torch.equal(bounds[0], torch.maximum(dropped, all_max[:, :, kg].max(dim=0).values))
    o_s = torch.empty((nq, k), dtype=torch.float32, device=eng.device)
    o_i = torch.empty((nq, k), dtype=torch.int64, device=eng.device)
    o_64 = torch.empty((nq, k), dtype=torch.float64, device=eng.device)
    status = torch.full((nq,), -1, dtype=torch.int32, device=eng.device)
    tau = eng.score_error_bound_any_plan(d)                    # what MatchPipeline certifies with: plan-independent
    assert tau >= max(eng.score_error_bound(nq, hi - lo, d, k) for lo, hi in shards)
    eng.topk_merge_packed(gathered, nq, k, out=(o_s, o_i), bound=bounds[0], tau=tau, scores_f64=o_64, status=status)
    assert torch.equal(o_i, want.idx) and torch.equal(o_s, want.scores) and torch.equal(o_64, want.scores_f64)
    assert int(status.min()) >= 0 and int(status.max()) <= 1
    # certificate of the merge == the definition, on the host
    kth = o_64[:, k - 1]
    assert torch.equal(status == 0, kth > bounds[0].double() + tau)
    assert kept <= nq * (kg + 4) and kept < parts * nq * kg * 0.5        # ~kg groups survive per query in total
    # the exhaustive round of the protocol, forced for EVERY query: each shard's exact list over all groups within tau of
    # the merged k-th score, merged again == the same result
    forced = torch.ones((nq,), dtype=torch.int32, device=eng.device)
    lower = kth.contiguous()
    for r, (lo, hi) in enumerate(shards):
        idx = gathered[r, :nq * k * 8].view(torch.int64).view(nq, k)
        sc = gathered[r, nq * k * 8:].view(torch.float64).view(nq, k)
        st = forced.clone()
        eng.exhaustive_topk(q, db[lo:hi], k, wss[r], lower, tau, st, sc, idx, row_offset=lo)
        assert int(st.min()) == 2 and int(st.max()) == 2
    eng.topk_merge_packed(gathered, nq, k, out=(o_s, o_i), scores_f64=o_64)
    assert torch.equal(o_i, want.idx) and torch.equal(o_64, want.scores_f64)


def test_host_staging_round_trips(eng):
    """dlc_host_to_device / dlc_device_to_host (pinned staging ring, host copy threads): byte-exact for sizes around the
    16 MiB piece and the 4-piece ring, several dtypes, non-default streams, back-to-back transfers that reuse the ring."""
    rng = np.random.RandomState(1)
    piece = 16 << 20
    for nbytes in (1, 7, 4096, piece - 1, piece, piece + 1, 3 * piece + 5, 4 * piece, 9 * piece + 123):
        a = rng.randint(0, 256, size=nbytes, dtype=np.uint8)
        t = eng.upload(a)
        a_copy = a.copy()
        a[:] = 0                                              # consumed: the caller may overwrite its array at once
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), a_copy)
        back = eng.download(t + 1)
        assert np.array_equal(back, a_copy + 1)
    s1 = torch.cuda.Stream(device=eng.device)
    for dt, shape in ((np.float64, (1000, 1681)), (np.float32, (3, 5, 7)), (np.int8, (1063, 2243)), (np.int64, (11,))):
        a = (rng.standard_normal(shape) * 50).astype(dt)
        with torch.cuda.stream(s1):
            t = eng.upload(a, stream=s1)
            u = t * 2
            got = eng.download(u, stream=s1)
        assert got.dtype == a.dtype and np.array_equal(got, a * 2)
    assert eng.upload(np.zeros(3, dtype=np.complex64)) is not None or True      # (dtype support is torch's)
    with pytest.raises(ValueError):
        eng.download(torch.zeros(8, device=eng.device), out=np.zeros(7, dtype=np.float32))


def test_transform_numpy_surface_is_chunked_and_bit_identical(dlc, eng):
    """The reference's contract (SDAV.py:293-302, cnn_vtl.py:130-133): ndarray in, ndarray out.  The host arrays go
    through Engine.run_chunked (upload / kernels / download of neighbouring chunks overlapped): same bits as the
    resident-tensor path whatever the chunking -- ragged last chunks, one chunk, chunks of one frame."""
    rng = np.random.RandomState(3)
    net = dlc.SDAV(seed=9)
    x = rng.uniform(0, 1, size=(37, 30, 1681))
    want = net.transform_tensor(torch.from_numpy(x).to(eng.device)).cpu().numpy()
    for cf in (128, 37, 36, 10, 1):
        got = net.transform(x, chunk_frames=cf)
        assert got.dtype == np.float64 and got.shape == (37 * 30, 2500) and np.array_equal(got, want), cf
    assert np.array_equal(net.transform(x.astype(np.float32).astype(np.float64)), net.transform(x.astype(np.float32)))
    assert net.transform(np.zeros((0, 30, 1681))).shape == (0, 2500)
    assert np.array_equal(net.transform(torch.from_numpy(x)), want)                      # a CPU tensor goes the old way
    with pytest.raises(ValueError):
        net.transform(np.zeros((2, 29, 1681)))
    cnn = dlc.CnnVtl(input_shape=[9, 192, 240, 3], seed=3, mask_seed=4)
    f8 = rng.randint(0, 256, size=(9, 192, 240, 3)).astype(np.uint8)
    want = cnn.transform_tensor(torch.from_numpy(f8).to(eng.device)).cpu().numpy()
    for cf in (None, 9, 4, 1):
        assert np.array_equal(cnn.transform(f8, chunk_frames=cf), want), cf
        assert np.array_equal(cnn.transform(f8.astype(np.float64), chunk_frames=cf), want), cf


def test_error_paths_raise_value_error(eng, dlc):
    """Bad arguments come back as negative status codes from the C ABI and surface as ValueError
    (the reference raises ValueError / validation errors); nothing is silently clamped."""
    f64 = lambda *s: torch.zeros(s, dtype=torch.float64, device=eng.device)
    with pytest.raises(ValueError):
        eng.gemm_bias_act(f64(4, 5), f64(6, 7))                                   # inner dimensions differ
    with pytest.raises(ValueError):
        eng.gemm_bias_act(f64(4, 5), f64(5, 7), bias=f64(3))                      # bias width
    with pytest.raises(ValueError):
        eng.gemm_bias_act(f64(4, 5).float(), f64(5, 7))                           # mixed dtypes
    with pytest.raises(ValueError):
        eng.sdav_similarity_matrix(f64(3, 65, 8), f64(8))                         # P > 64 patches
    with pytest.raises(ValueError):
        dlc.SimilarityCalculator(np.zeros((3, 30)))                               # dataset must be [N,P,H]
    with pytest.raises(ValueError):
        dlc.DistanceCalculator.distance_matrix(np.zeros((3,), dtype=np.int8))
    with pytest.raises(ValueError):
        dlc.CnnVtl(input_shape=[1, 224, 224, 1])
    with pytest.raises(ValueError):
        dlc.CnnVtl(compress_factor=120.0)
    with pytest.raises(ValueError):
        eng.groups_per_query(0)
    q = torch.zeros((4, 64), dtype=torch.bfloat16, device=eng.device)
    ws = torch.empty(eng.topk_workspace_bytes(4, 4, 64, 2), dtype=torch.uint8, device=eng.device)
    with pytest.raises(L_ERRORS):
        eng.score_groups(q, q, 2, ws[:128])                                       # workspace too small
    import ctypes as C
    buf = torch.empty(1024, dtype=torch.uint8, device=eng.device)
    rc = eng.lib.dlc_set_scratch(eng.ctx, C.c_void_p(buf.data_ptr() + 8), 512)    # not 256-byte aligned
    assert rc == dlc._lib.DLC_ERR_BAD_ARG and b"aligned" in eng.lib.dlc_last_error(eng.ctx)
    long_rows = torch.zeros((8, 20032), dtype=torch.bfloat16, device=eng.device)
    s_out = torch.empty((8, 8), dtype=torch.float32, device=eng.device)
    assert eng.lib.dlc_cosine_scores_workspace_bytes(8, 8, 20032) > 0
    rc = eng.lib.dlc_cosine_scores(eng.ctx, dlc._lib.DLC_BF16, C.c_void_p(long_rows.data_ptr()), 8, 20032,
                                   C.c_void_p(long_rows.data_ptr()), 8, 20032, 20032, C.c_void_p(s_out.data_ptr()), 8,
                                   None, 0, None)
    assert rc == dlc._lib.DLC_ERR_WORKSPACE                                       # split-K shape without its workspace
    assert dlc.DistanceCalculator.calculate_distance([], []) == 0                 # zip of empties (reference: 0)
    assert dlc.DistanceCalculator.calculate_distance([1, 2, 3], [1]) == 0         # zip stops at the shorter one


from deeploopcloser_amd._lib import DlcError as _DlcError   # noqa: E402
L_ERRORS = (_DlcError, ValueError)


def test_row_stride_bound_and_buffer_validation(eng, dlc):
    """The score GEMM addresses a lane's bytes as a 32-bit offset inside its 256-row tile: row strides for which
    255 rows would not fit below 4 GiB are refused (DLC_ERR_BAD_SHAPE), not wrapped.  Caller-supplied output
    buffers of the wrong shape / dtype / layout are refused before any pointer reaches a kernel."""
    import ctypes as C
    L = dlc._lib
    q = torch.zeros((4, 64), dtype=torch.bfloat16, device=eng.device)
    s = torch.empty((4, 2), dtype=torch.float32, device=eng.device)
    i = torch.empty((4, 2), dtype=torch.int64, device=eng.device)
    need = eng.lib.dlc_cosine_topk_workspace_bytes(4, 4, 64, 2)
    ws = torch.empty(need, dtype=torch.uint8, device=eng.device)
    big = (1 << 32) // (2 * 255) // 8 * 8 + 8                       # elements: 255 rows x 2 bytes >= 4 GiB
    for ldq, lddb in ((64, big), (big, 64)):
        rc = eng.lib.dlc_cosine_topk(eng.ctx, L.DLC_BF16, C.c_void_p(q.data_ptr()), 4, ldq, C.c_void_p(q.data_ptr()), 1, lddb,
                                     64, 2, 0, C.c_void_p(s.data_ptr()), None, C.c_void_p(i.data_ptr()), None, None,
                                     C.c_void_p(ws.data_ptr()), ws.numel(), None)
        assert rc == L.DLC_ERR_BAD_SHAPE and b"stride" in eng.lib.dlc_last_error(eng.ctx)
    ok = (1 << 32) // (2 * 255) // 8 * 8 - 64                       # just below the bound: accepted (one row: the stride is unused)
    rc = eng.lib.dlc_cosine_topk(eng.ctx, L.DLC_BF16, C.c_void_p(q.data_ptr()), 1, ok, C.c_void_p(q.data_ptr()), 1, ok, 64, 1, 0,
                                 C.c_void_p(s.data_ptr()), None, C.c_void_p(i.data_ptr()), None, None,
                                 C.c_void_p(ws.data_ptr()), ws.numel(), None)
    torch.cuda.synchronize()
    assert rc == L.DLC_OK
    for bad in ((s[:, :1], i), (s, i.to(torch.int32)), (s.t().contiguous().t(), i), (s.cpu(), i)):
        with pytest.raises(ValueError):
            eng.match_topk(q, q, 2, out=bad)
    with pytest.raises(ValueError):
        eng.select_topk(q, q, 2, ws, s[:2], i)
    with pytest.raises(ValueError):
        eng.topk_merge(torch.zeros((2, 4, 2), dtype=torch.float64, device=eng.device),
                       torch.zeros((2, 4, 2), dtype=torch.int64, device=eng.device), out=(s[:3], i))
    with pytest.raises(ValueError):                                # the order is decided in fp64: fp32 parts are refused
        eng.topk_merge(torch.zeros((2, 4, 2), device=eng.device), torch.zeros((2, 4, 2), dtype=torch.int64, device=eng.device))
    with pytest.raises(ValueError):
        eng.gemm_bias_act(torch.zeros((4, 5), dtype=torch.float64, device=eng.device),
                          torch.zeros((5, 7), dtype=torch.float64, device=eng.device),
                          out=torch.zeros((4, 6), dtype=torch.float64, device=eng.device))


def test_create_rejects_a_device_that_is_not_there(dlc):
    """dlc_create(device, &ctx) on an index past the visible GPUs (what a mis-set LOCAL_RANK hands it) or a negative one:
    DLC_ERR_BAD_ARG, a null context, no HIP state touched -- and the Python engine raises instead of keeping a half-made
    handle.  The one-GPU half of the per-device path (the two-GPU half below needs a second GPU)."""
    import ctypes as C
    from deeploopcloser_amd import _lib as L
    lib = L.load()
    for bad in (torch.cuda.device_count(), torch.cuda.device_count() + 7, -1):
        ctx = C.c_void_p(0xdead)
        assert lib.dlc_create(bad, C.byref(ctx)) == L.DLC_ERR_BAD_ARG and not ctx.value
    assert lib.dlc_create(0, None) == L.DLC_ERR_BAD_ARG
    from deeploopcloser_amd.engine import Engine
    with pytest.raises(L.DlcError):
        Engine(torch.cuda.device_count())
    # ... and the engine of device 0 is untouched by the refusals
    e = dlc.default_engine(0)
    assert float(e.normalize(torch.ones((2, 64), device=e.device), "bf16").float().norm(dim=1).max()) == pytest.approx(1.0, abs=1e-2)


def test_two_contexts_two_devices(dlc):
    """One context per GPU: the kernels' dynamic-LDS limits are set per DEVICE (the flags live in the context), so a
    second context on another GPU of the same process runs the 128 KiB score GEMM too.  Needs two visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the per-device attribute path needs two")
    rng = np.random.RandomState(1)
    x = rng.standard_normal((3000, 256)).astype(np.float32)
    res = []
    for dev in (0, 1):
        e = dlc.default_engine(dev)
        with torch.cuda.device(dev):
            st = e.normalize(torch.from_numpy(x).to(e.device), "bf16")
            s_, i_ = e.match_topk(st[:300], st, 5)
            h = e.gemm_bias_act(torch.ones((200, 300), dtype=torch.float64, device=e.device),
                                torch.ones((300, 130), dtype=torch.float64, device=e.device))
            torch.cuda.synchronize(dev)
            assert float(h.min()) == 300.0 == float(h.max())
            res.append((s_.cpu(), i_.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_normalize_one_pass_forms_vs_reference_rounding(eng, dlc):
    """dlc_l2_normalize_rows: the wave-per-row and workgroup-per-row register forms and the multi-pass fall-back
    (wide rows, unaligned rows) against the same fp64 formula, for fp32 / fp64 sources and both stored types."""
    rng = np.random.RandomState(5)
    for d in (64, 100, 1000, 4096, 4097, 5000, 16384, 16400, 75000):
        for src in (np.float32, np.float64):
            x = rng.standard_normal((7, d)).astype(src)
            for center in (False, True):
                for dt, tdt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
                    got = eng.normalize(torch.from_numpy(x).to(eng.device), dt, center)
                    assert got.shape == (7, (d + 63) // 64 * 64) and got.dtype == tdt
                    y = x.astype(np.float64) - (x.astype(np.float64).mean(1, keepdims=True) if center else 0.0)
                    ref = torch.from_numpy((y / np.linalg.norm(y, axis=1, keepdims=True)).astype(np.float32)).to(tdt).float().numpy()
                    g = got.float().cpu().numpy()
                    assert np.all(g[:, d:] == 0)
                    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
                    assert np.abs(g[:, :d] - ref).max() <= ulp * np.abs(ref).max() and (g[:, :d] == ref).mean() > 0.99, (d, src, center, dt)
    # a strided (unaligned) source view takes the multi-pass kernel: same values as the aligned copy
    x = torch.from_numpy(rng.standard_normal((5, 1001)).astype(np.float32)).to(eng.device)
    a = eng.normalize(x[:, 1:].contiguous(), "bf16")
    wide = torch.zeros((5, 1003), dtype=torch.float32, device=eng.device)
    wide[:, 3:] = x[:, 1:]
    rc_view = wide[:, 3:]
    out = torch.empty_like(a)
    eng._check(eng.lib.dlc_l2_normalize_rows(eng.ctx, dlc._lib.DLC_F32, rc_view.data_ptr(), 5, 1000, rc_view.stride(0), 0,
                                              dlc._lib.DLC_BF16, out.data_ptr(), out.stride(0), None))
    torch.cuda.synchronize()
    assert (a.float() - out.float()).abs().max().item() <= 2.0 ** -8


@pytest.mark.parametrize("m,n,k,blayout", [(33000, 770, 338, "kn"), (65537, 256, 64, "kn"), (40000, 514, 1002, "nk"),
                                           (70000, 128, 2500, "nk"), (257 * 128, 1026, 70, "kn")])
def test_gemm_dma_kernel_edges(eng, m, n, k, blayout):
    """Launches large enough for the LDS-DMA fp64 kernel (>= 512 tiles of 256 x 128, even K / N / strides), with every
    edge it clamps or masks: M, N not multiples of the tile, a K tail that ends inside a 16-byte piece's pair and one
    that ends inside a K tile, both B layouts -- against a torch fp64 product (the checker), and bit-identical to the
    register-staged kernel, reached by giving A an odd row stride (8-byte-aligned rows cannot be DMA'd)."""
    from deeploopcloser_amd import _lib as L
    g = torch.Generator(device=eng.device)
    g.manual_seed(m + n + k)
    a = torch.randn((m, k), generator=g, device=eng.device, dtype=torch.float64) / k ** 0.5
    b = torch.randn((k, n) if blayout == "kn" else (n, k), generator=g, device=eng.device, dtype=torch.float64)
    bias = torch.randn((n,), generator=g, device=eng.device, dtype=torch.float64)
    lay = L.DLC_B_KN if blayout == "kn" else L.DLC_B_NK
    got = eng.gemm_bias_act(a, b, bias, act=L.DLC_ACT_SIGMOID, blayout=lay)
    ref = torch.sigmoid(a @ (b if blayout == "kn" else b.T) + bias)
    assert float((got - ref).abs().max()) < 1e-12
    # the same operands with an odd leading dimension of A: the register-staged kernel, same k order -> same bits
    import ctypes as C
    wide = torch.zeros((m, k + 1), dtype=torch.float64, device=eng.device)
    wide[:, :k] = a
    out = torch.empty((m, n), dtype=torch.float64, device=eng.device)
    eng._check(eng.lib.dlc_gemm_bias_act(eng.ctx, L.DLC_F64, lay, L.DLC_ACT_SIGMOID, m, n, k, wide.data_ptr(), k + 1,
                                          b.data_ptr(), b.stride(0), bias.data_ptr(), out.data_ptr(), n, None))
    torch.cuda.synchronize()
    assert torch.equal(out, got)


def test_gemm_dma_small_launch_fuzz(eng):
    """The LDS-DMA kernel is taken from 16 tiles on: random small / medium shapes (edge blocks numbered rows-first,
    blocks shrunk for the shader engines, N <= 96 on the 96-column tile, K tails, both B layouts, all activations)
    against torch fp64 and, bit for bit, against the register-staged kernel (odd row stride of A)."""
    from deeploopcloser_amd import _lib as L
    rng = np.random.RandomState(77)
    g = torch.Generator(device=eng.device); g.manual_seed(77)
    for _ in range(36):
        m = int(rng.randint(1, 6000))                            # any row count: the 128-row tile takes small ones
        n = int(rng.choice([2, 34, 66, 96, 98, 128, 130, 256, 300, 384, 640, 1024])) if rng.rand() < 0.7 else 2 * int(rng.randint(1, 400))
        k = 2 * int(rng.randint(32, 400))
        lay = L.DLC_B_KN if rng.rand() < 0.5 else L.DLC_B_NK
        act = int(rng.randint(0, 3))
        a = torch.randn((m, k), generator=g, device=eng.device, dtype=torch.float64) / k ** 0.5
        b = torch.randn((k, n) if lay == L.DLC_B_KN else (n, k), generator=g, device=eng.device, dtype=torch.float64)
        bias = torch.randn((n,), generator=g, device=eng.device, dtype=torch.float64)
        got = eng.gemm_bias_act(a, b, bias, act=act, blayout=lay)
        z = a @ (b if lay == L.DLC_B_KN else b.T) + bias
        ref = torch.sigmoid(z) if act == L.DLC_ACT_SIGMOID else (torch.relu(z) if act == L.DLC_ACT_RELU else z)
        assert float((got - ref).abs().max()) < 1e-11, (m, n, k, lay, act)
        wide = torch.zeros((m, k + 1), dtype=torch.float64, device=eng.device)
        wide[:, :k] = a
        out = torch.empty((m, n), dtype=torch.float64, device=eng.device)
        eng._check(eng.lib.dlc_gemm_bias_act(eng.ctx, L.DLC_F64, lay, act, m, n, k, wide.data_ptr(), k + 1,
                                              b.data_ptr(), b.stride(0), bias.data_ptr(), out.data_ptr(), n, None))
        torch.cuda.synchronize()
        assert torch.equal(out, got), (m, n, k, lay, act)


def test_space_to_depth_and_conv1_equivalence(eng, dlc):
    """dlc_space_to_depth_nhwc_f64 against the NumPy reshape / transpose, and CnnVtl's conv1 through it (a 3x3
    convolution over 48 channels) against the oracle's 11x11 / 4 convolution and against the direct call."""
    from oracle import cnn_vtl as ocnn
    from deeploopcloser_amd import _lib as L
    rng = np.random.RandomState(2)
    for (n, h, w, c, s) in [(3, 8, 12, 3, 4), (2, 6, 6, 5, 2), (1, 192, 240, 3, 4)]:
        x = rng.standard_normal((n, h, w, c))
        got = eng.space_to_depth(torch.from_numpy(x).to(eng.device), s).cpu().numpy()
        ref = x.reshape(n, h // s, s, w // s, s, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, h // s, w // s, s * s * c)
        assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        eng.space_to_depth(torch.zeros((1, 7, 8, 3), dtype=torch.float64, device=eng.device), 4)
    net = dlc.CnnVtl(input_shape=[2, 192, 240, 3], seed=5, mask_seed=9)
    assert 0 in net._s2d and net._s2d[0][:3] == (4, 3, 3)
    x = rng.randint(0, 256, size=(2, 192, 240, 3)).astype(np.float64)
    outs = net._features(torch.from_numpy(x).to(eng.device))
    ws, bs = ocnn.init_weights(5)
    ref1 = ocnn.conv2d_nhwc(x, ws[0], bs[0], 4, "VALID", True)
    assert outs[0].shape == (2, 46, 58, 96) and np.abs(outs[0].cpu().numpy() - ref1).max() < 1e-9 * np.abs(ref1).max()
    direct = eng.conv2d(torch.from_numpy(x).to(eng.device), net._w[0], net._b[0], 11, 11, 4, 0, 0, 46, 58, L.DLC_ACT_RELU)
    assert float((direct - outs[0]).abs().max()) < 1e-9 * np.abs(ref1).max()


def test_engine_workspaces_are_bounded_per_name(eng):
    """Engine.workspace keeps one tensor per (name, stream); a caller that rotates streams must not accumulate them."""
    streams = [torch.cuda.Stream(device=eng.device) for _ in range(12)]
    q = torch.zeros((4, 64), dtype=torch.bfloat16, device=eng.device)
    for s in streams:
        with torch.cuda.stream(s):
            eng.match_topk(q, q, 2)
    torch.cuda.synchronize()
    assert sum(1 for k in eng._ws if k[0] == "topk") <= eng.WORKSPACE_STREAMS


def test_similarity_range_from_the_distinctive_pass(eng, dlc):
    """dlc_sdav_distinctive_score can leave every column's extremes and the NaN / infinity flag for
    dlc_sdav_similarity_matrix (the filter form then reads the descriptors once less): same matrix as without; the flag
    routes a dataset with a NaN to the fp64 form as before; the class uses it for its own dataset only."""
    g = torch.Generator(device=eng.device); g.manual_seed(12)
    ds = 3.0 * torch.randn((40, 30, 250), generator=g, device=eng.device, dtype=torch.float64) - 1.0
    score, rng = eng.distinctive_score(ds, 0.5, 0.2, with_range=True)
    assert torch.equal(score, eng.distinctive_score(ds, 0.5, 0.2))
    cols = ds.reshape(-1, 250).cpu().numpy()
    key = lambda v: int(np.frombuffer(np.float64(v).tobytes(), dtype=np.uint64)[0])
    okey = lambda v: (~key(v)) & (2 ** 64 - 1) if key(v) >> 63 else key(v) | (1 << 63)
    got = [int(x) & (2 ** 64 - 1) for x in rng.cpu().tolist()]
    assert len(got) == 3 + 2 * 250 and got[2] == 0
    assert got[3:253] == [okey(v) for v in cols.min(0)] and got[253:] == [okey(v) for v in cols.max(0)]
    a = [t.clone() for t in eng.sdav_similarity_matrix(ds, score, 10.0, -10.0)]
    b = eng.sdav_similarity_matrix(ds, score, 10.0, -10.0, range=rng)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    bad = ds.clone(); bad[3, 4, 5] = float("inf")
    score, rng = eng.distinctive_score(bad, 0.5, 0.2, with_range=True)
    assert int(rng[2]) == 1
    a = [t.clone() for t in eng.sdav_similarity_matrix(bad, score, 10.0, -10.0)]
    b = eng.sdav_similarity_matrix(bad, score, 10.0, -10.0, range=rng)
    assert torch.equal(torch.nan_to_num(a[0]), torch.nan_to_num(b[0])) and torch.equal(a[1], b[1])
    calc = dlc.SimilarityCalculator(ds.cpu().numpy())
    assert np.array_equal(calc.similarity_matrix(), calc.similarity_matrix(ds.cpu().numpy()))      # own dataset (range kept) == any descriptors
