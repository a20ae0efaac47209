"""NumPy restatement of the SDAV similarity (src/sdav/similarity/
SimilarityCalculator.py:1-49) and of the all-vs-all loop of
src/sdav/create_similarity_matrix.py:29-38.  Test infrastructure only.
PINNED against the reference module's own outputs (tests/golden/similarity.npz)."""
import numpy as np

INT64_MIN = np.iinfo(np.int64).min


def average_response(dataset):
    """_average_response (SimilarityCalculator.py:20-23)."""
    x0 = dataset.shape[0] * dataset.shape[1]
    return np.average(dataset.reshape(x0, dataset.shape[2]), axis=0)


def distinctive_score(avg, mu=0.5, sigma=0.2):
    """_distinctive_score (SimilarityCalculator.py:25-27)."""
    return np.exp(-((avg - mu) ** 2) / (2 * sigma ** 2))


def match_features(h1, h2):
    """_match_features (SimilarityCalculator.py:30-37): for each row of h1 the
    index of the nearest (L2) row of h2, first minimum wins (np.argmin)."""
    idx = np.empty(h1.shape[0], dtype=np.int64)
    for i, mi in enumerate(h1):
        idx[i] = np.argmin(np.linalg.norm(h2 - mi, axis=1))
    return idx


def weighted_distances(h1, h2, idx, score):
    """_compute_weighted_distances (SimilarityCalculator.py:40-45):
    |dot(score, m_i - h2[j*])| (norm of a scalar = abs)."""
    return np.array([np.linalg.norm(np.matmul(score, h1[i] - h2[j])) for i, j in enumerate(idx)])


def similarity_score(dataset, h1, h2, mu=0.5, sigma=0.2, a=10, b=-10):
    """similarity_score (SimilarityCalculator.py:12-17,47-49). d == 0 -> +inf."""
    s = distinctive_score(average_response(dataset), mu, sigma)
    idx = match_features(h1, h2)
    d = weighted_distances(h1, h2, idx, s)
    with np.errstate(divide="ignore"):
        return np.sum(a + b * np.log(d))


def truncate_to_int64(v):
    """Store of a python/NumPy float into the int64 matrix of
    create_similarity_matrix.py:31,36-37: C cast, truncation toward zero; a
    non-finite value becomes INT64_MIN (x86 cvttsd2si, which the pinned
    numpy 1.15 used; newer NumPy raises instead)."""
    v = np.asarray(v, dtype=np.float64)
    out = np.full(v.shape, INT64_MIN, dtype=np.int64)
    ok = np.isfinite(v) & (np.abs(v) < 2.0 ** 63)
    out[ok] = np.trunc(v[ok]).astype(np.int64)
    return out


def similarity_matrix_f64(descriptors, mu=0.5, sigma=0.2, a=10, b=-10):
    """All-vs-all loop (create_similarity_matrix.py:29-38) before the int64
    store: only i<j is computed (score(h_i, h_j)), mirrored to [j,i]; the
    diagonal stays -1.  average/distinctive score hoisted (same value for
    every pair)."""
    descriptors = np.asarray(descriptors, dtype=np.float64)
    n = descriptors.shape[0]
    s = distinctive_score(average_response(descriptors), mu, sigma)
    out = np.full((n, n), -1.0)
    for i in range(n):
        for j in range(i + 1, n):
            idx = match_features(descriptors[i], descriptors[j])
            d = weighted_distances(descriptors[i], descriptors[j], idx, s)
            with np.errstate(divide="ignore"):
                v = np.sum(a + b * np.log(d))
            out[i, j] = v
            out[j, i] = v
    return out


def similarity_matrix(descriptors, **kw):
    """int64 matrix exactly as create_similarity_matrix.py:31-38 leaves it."""
    return truncate_to_int64(similarity_matrix_f64(descriptors, **kw))
