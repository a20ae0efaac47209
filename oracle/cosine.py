"""fp64 oracle for the cosine-similarity + top-k match.

NOT IN THE REFERENCE: this function is defined by BASELINE.json's north_star
 -- "parity unpinned".  Definition used by the
whole build: rows are taken AS STORED (already L2-normalised and rounded to
bf16/fp16 by the caller), S = Q . DB^T in fp64, per query the k largest scores,
descending, ties broken toward the LOWER database index.  Test infrastructure
only."""
import numpy as np


def l2_normalize(x, center=False):
    x = np.asarray(x, dtype=np.float64)
    if center:
        x = x - x.mean(axis=1, keepdims=True)
    n = np.linalg.norm(x, axis=1, keepdims=True)
    n[n == 0] = 1.0
    return x / n


def scores(q, db):
    return np.asarray(q, dtype=np.float64) @ np.asarray(db, dtype=np.float64).T


_TIE_Q = 2.0 ** 40


def order_key(s):
    """Ordering key of fp64 scores: quantised to 2^-40 so that BLAS rounding noise
    (identical database rows can get dot products 1 ulp apart, depending on
    where they sit in a block) cannot break an exact tie the wrong way.  2^-40
    is ~5 orders below fp32 resolution and ~4 above fp64 noise for |s| <= 1."""
    return np.round(np.asarray(s, dtype=np.float64) * _TIE_Q) / _TIE_Q


def topk_from_scores(s, k, row_offset=0):
    """Top-k per row: score descending, ties -> lower index."""
    q, n = s.shape
    s_raw, s = s, order_key(s)
    k = min(k, n)
    out_s = np.empty((q, k), dtype=np.float64)
    out_i = np.empty((q, k), dtype=np.int64)
    for r in range(q):
        row = s[r]
        if n > 4 * k:
            cand = np.argpartition(-row, k - 1)[:k]
            thr = row[cand].min()
            cand = np.nonzero(row >= thr)[0]          # keep every tie of the k-th score
        else:
            cand = np.arange(n)
        order = np.lexsort((cand, -row[cand]))[:k]
        out_i[r] = cand[order] + row_offset
        out_s[r] = s_raw[r][cand[order]]
    return out_s, out_i


def cosine_topk(q, db, k, row_offset=0, block=65536, dtype=np.float64):
    """Blocked exact top-k so that a 100k-row oracle run stays in memory.  dtype: the matmul's
    arithmetic (float64 = the oracle; float32 only for bench.py's fp32 CPU timing)."""
    q = np.asarray(q, dtype=dtype)
    n = db.shape[0]
    best_s = np.empty((q.shape[0], 0))
    best_i = np.empty((q.shape[0], 0), dtype=np.int64)
    for lo in range(0, n, block):
        s = q @ np.asarray(db[lo:lo + block], dtype=dtype).T
        bs, bi = topk_from_scores(s, k, row_offset + lo)
        best_s, best_i = merge_topk(np.concatenate([best_s, bs], 1), np.concatenate([best_i, bi], 1), k)
    return best_s, best_i


def merge_topk(cand_s, cand_i, k):
    """k-way merge of per-shard candidates [q, parts*k] with the same ordering
    rule (score desc, lower GLOBAL index first)."""
    q = cand_s.shape[0]
    k = min(k, cand_s.shape[1])
    out_s = np.empty((q, k), dtype=cand_s.dtype)
    out_i = np.empty((q, k), dtype=np.int64)
    key = order_key(cand_s)
    for r in range(q):
        order = np.lexsort((cand_i[r], -key[r]))[:k]
        out_s[r] = cand_s[r][order]
        out_i[r] = cand_i[r][order]
    return out_s, out_i
