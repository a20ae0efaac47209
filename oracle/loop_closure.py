"""fp64 oracle of the streaming loop-closure query (test infrastructure only).

NOT IN THE REFERENCE (SURVEY.md section 8f-4 asks for "a streaming loop-closure query CLI
over the top-k engine"); definition used by the build: key-frame g (0-based arrival order)
is matched against the key-frames that are MORE than `exclusion` frames older, i.e. ids
0 .. g-exclusion-1, with the cosine top-k rule of oracle/cosine.py (score descending, ties to
the lower id) on the rows AS STORED; slots without an eligible key-frame hold id -1 and
score -inf.  Plain per-frame loop, independent of how the product batches the frames."""
import numpy as np

from . import cosine


def stream_topk(stored_rows, k, exclusion):
    """stored_rows [T, D] (values as stored, any float dtype) -> (scores [T,k] f64, idx [T,k] i64)."""
    rows = np.asarray(stored_rows, dtype=np.float64)
    t = rows.shape[0]
    out_s = np.full((t, k), -np.inf)
    out_i = np.full((t, k), -1, dtype=np.int64)
    for g in range(t):
        n = g - exclusion
        if n <= 0:
            continue
        s, i = cosine.topk_from_scores(rows[g:g + 1] @ rows[:n].T, k)
        out_s[g, :s.shape[1]] = s[0]
        out_i[g, :i.shape[1]] = i[0]
    return out_s, out_i
