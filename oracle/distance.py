"""NumPy restatement of the cnn_vtl descriptor distance
(src/cnn_vtl/similarity/DistanceCalculator.py:1-12) and of the all-vs-all loop
of src/cnn_vtl/create_distance_matrix.py:30-36.  Test infrastructure only.
PINNED against the reference module's own outputs (tests/golden/distance.npz)."""
import numpy as np

# popcount(|x|) for x in -128..127: bin() of a negative NumPy int8 prints
# '-0b<magnitude>' so the reference counts the bits of the MAGNITUDE
# (DistanceCalculator.py:4-5); |-128| = 128 -> 1 bit.
_POPABS = np.array([bin(abs(v)).count("1") for v in range(-128, 128)], dtype=np.int64)


def bitwise_diff(a, b):
    """_bitwise_diff (DistanceCalculator.py:4-5) on int8 arrays, elementwise."""
    x = np.bitwise_xor(np.asarray(a, dtype=np.int8), np.asarray(b, dtype=np.int8))
    return _POPABS[x.astype(np.int64) + 128]


def calculate_distance(d1, d2):
    """DistanceCalculator.calculate_distance (DistanceCalculator.py:10-12)."""
    return np.int64(np.sum(bitwise_diff(d1, d2)))


def distance_matrix(desc):
    """Full N x N loop incl. the diagonal (create_distance_matrix.py:30-36)."""
    desc = np.asarray(desc, dtype=np.int8)
    n = desc.shape[0]
    out = np.empty((n, n), dtype=np.int64)
    for i in range(n):
        out[i] = bitwise_diff(desc[i][None, :], desc).sum(axis=1)
    return out
