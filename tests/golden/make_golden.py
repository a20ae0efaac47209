#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ (run in the BUILD container only).

The reference's NumPy-only modules (SimilarityCalculator, DistanceCalculator,
MathUtils) import and run here, so their OUTPUTS on seeded inputs are captured
as data.  The TensorFlow / OpenCV modules do not import (ModuleNotFoundError),
so the encoder has no reference-generated vector: its only upstream pin is the
literal of test/TensorflowWrapperTest.py:12-14, copied as data.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing here is needed (or present) on the GPU box; the tests read only the
.npz files this writes.
"""
import os
import sys

import numpy as np

REF = os.environ.get("DLC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from src.sdav.similarity.SimilarityCalculator import SimilarityCalculator  # noqa: E402
from src.cnn_vtl.similarity.DistanceCalculator import DistanceCalculator   # noqa: E402
from src.utils.MathUtils import MathUtils                                   # noqa: E402


def similarity_cases():
    out = {}
    cases = [("n6_h8", 6, 30, 8, 11, {}), ("n6_h64", 6, 30, 64, 12, {}),
             ("n4_h2500", 4, 30, 2500, 13, {}),
             ("n5_h32_params", 5, 30, 32, 14, dict(mu=0.4, sigma=0.3, a=7, b=-3)),
             ("n5_p7_h16", 5, 7, 16, 15, {})]
    for name, n, p, h, seed, kw in cases:
        rng = np.random.RandomState(seed)
        ds = rng.uniform(0.0, 1.0, size=(n, p, h))
        if name == "n6_h64":
            ds[3] = ds[1]                      # identical frames -> d == 0 -> +inf
            ds[4, 5] = ds[2, 9]                # one identical patch across frames
        calc = SimilarityCalculator(ds, **kw)
        avg = SimilarityCalculator._average_response(ds)
        dscore = calc._distinctive_score(avg)
        scores = np.zeros((n, n))
        argmins = np.zeros((n, n, p), dtype=np.int64)
        wdist = np.zeros((n, n, p))
        with np.errstate(divide="ignore"):
            for i in range(n):
                for j in range(n):
                    matched = SimilarityCalculator._match_features(ds[i], ds[j])
                    for r, (mi, mj) in enumerate(matched):
                        argmins[i, j, r] = int(np.nonzero((ds[j] == mj).all(axis=1))[0][0])
                    wdist[i, j] = SimilarityCalculator._compute_weighted_distances(matched, dscore)
                    scores[i, j] = calc.similarity_score(ds[i], ds[j])
        if ds.nbytes <= 200_000:
            out[name + "/dataset"] = ds
        else:                                  # keep the fixture small: RandomState streams are frozen
            out[name + "/dataset_seed_uniform01"] = np.array([seed, n, p, h], dtype=np.int64)
        out[name + "/params"] = np.array([kw.get("mu", 0.5), kw.get("sigma", 0.2), kw.get("a", 10), kw.get("b", -10)],
                                         dtype=np.float64)
        out[name + "/average_response"] = avg
        out[name + "/distinctive_score"] = dscore
        out[name + "/argmin"] = argmins
        out[name + "/weighted_distances"] = wdist
        out[name + "/scores"] = scores
    return out


def distance_cases():
    out = {}
    a = np.array([-1, 5, -56, 127, -128], dtype=np.int8)
    b = np.array([0, 5, 0, -128, 127], dtype=np.int8)
    out["probe/a"], out["probe/b"] = a, b
    out["probe/distance"] = np.int64(DistanceCalculator.calculate_distance(a, b))
    # every int8 XOR result once: a ^ 0 = a
    allv = np.arange(-128, 128).astype(np.int8)
    out["all/a"] = allv
    out["all/per_element"] = np.array([DistanceCalculator.calculate_distance([v], [np.int8(0)]) for v in allv],
                                      dtype=np.int64)
    rng = np.random.RandomState(21)
    for name, n, d in (("n7_d2243", 7, 2243), ("n9_d37", 9, 37), ("n3_d1", 3, 1)):
        desc = rng.randint(-128, 128, size=(n, d)).astype(np.int8)
        m = np.empty((n, n), dtype=np.int64)
        for i in range(n):
            for j in range(n):
                m[i, j] = DistanceCalculator.calculate_distance(desc[i], desc[j])
        out[name + "/desc"] = desc
        out[name + "/matrix"] = m
    return out


def mathutils_cases():
    vals = np.array([279936, 173056, 55296, 36864, 256128, 157696, 49920, 33280, 1, 1000, 12345], dtype=np.int64)
    pct = 99.59
    return {"values": vals, "compression": np.float64(pct),
            "sizes": np.array([MathUtils.compressed_size(int(v), pct) for v in vals], dtype=np.int64),
            "sizes_50": np.array([MathUtils.compressed_size(int(v), 50.0) for v in vals], dtype=np.int64)}


def tensorwrapper_literal():
    # Data literal of test/TensorflowWrapperTest.py:12-14 (x, w, expected).
    return {"x": np.array([[[1, 2], [3, 4]], [[5, 6], [7, 8]], [[9, 10], [11, 12]]], dtype=np.float64),
            "w": np.array([[2, 2], [2, 2]], dtype=np.float64),
            "expected": np.array([[[6, 6], [14, 14]], [[22, 22], [30, 30]], [[38, 38], [46, 46]]], dtype=np.float64)}


def main():
    np.savez_compressed(os.path.join(HERE, "similarity.npz"), **similarity_cases())
    np.savez_compressed(os.path.join(HERE, "distance.npz"), **distance_cases())
    np.savez_compressed(os.path.join(HERE, "mathutils.npz"), **mathutils_cases())
    np.savez_compressed(os.path.join(HERE, "tensorwrapper_test_example.npz"), **tensorwrapper_literal())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
