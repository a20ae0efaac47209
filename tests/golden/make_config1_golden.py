#!/usr/bin/env python3
"""Capture tests/golden/config1.npz (run in the BUILD container only).

BASELINE configs[0] at size: the 20 frames of the reference's datasets/test -> oracle patches -> oracle
SDAV descriptors (TensorFlow is not installable, so the ENCODER stays parity-unpinned) -> the
REFERENCE's own SimilarityCalculator (importable here: NumPy only) driven with the loop shape of
src/sdav/create_similarity_matrix.py:29-38; and oracle CnnVtl descriptors -> the REFERENCE's DistanceCalculator
in the loop of src/cnn_vtl/create_distance_matrix.py:30-36.  The matrices it returns for these descriptors are the
golden data; tests/test_config1.py regenerates the descriptors with the oracle and checks the
oracle's and the GPU's matrices against them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_config1_golden.py
"""
import os
import sys

import numpy as np

REF = os.environ.get("DLC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from src.sdav.similarity.SimilarityCalculator import SimilarityCalculator  # noqa: E402
from src.cnn_vtl.similarity.DistanceCalculator import DistanceCalculator   # noqa: E402
import config1_common as c1                                                 # noqa: E402


def main():
    paths = c1.frame_paths()
    assert len(paths) == 20
    x = c1.oracle_patches(paths)
    out = {"frames": np.array([os.path.basename(p) for p in paths])}
    for scale in ("reference", "fan_in"):
        h = c1.oracle_descriptors(x, scale)
        ds = h.reshape(20, 30, 2500)
        calc = SimilarityCalculator(ds)
        m = np.full([20, 20], -1.0)
        for i in range(20):                                 # create_similarity_matrix.py:34-38 (i < j, mirrored)
            for j in range(i + 1, 20):
                with np.errstate(divide="ignore"):
                    m[i, j] = m[j, i] = calc.similarity_score(ds[i], ds[j])
        out["similarity_f64_" + scale] = m
        out["descriptor_sum_" + scale] = np.array(h.sum())
        print(scale, "finite pairs:", int(np.isfinite(m).sum() - 20) // 2, "of 190; descriptor sum", h.sum())
    # configs[2] at the same size: oracle CnnVtl descriptors of the 20 frames -> the REFERENCE's DistanceCalculator in
    # the full N x N loop of src/cnn_vtl/create_distance_matrix.py:30-36 (diagonal and both triangles evaluated)
    d8 = c1.oracle_cnn_descriptors(paths)
    dm = np.zeros([20, 20], dtype=np.int64)
    for i in range(20):
        for j in range(20):
            dm[i, j] = DistanceCalculator.calculate_distance(d8[i], d8[j])
    out["distance_i64"] = dm
    out["cnn_descriptor_sum"] = np.array(int(d8.astype(np.int64).sum()))
    out["cnn_descriptor_width"] = np.array(d8.shape[1])
    print("cnn_vtl descriptors", d8.shape, "distance matrix max", dm.max())
    np.savez_compressed(os.path.join(HERE, "config1.npz"), **out)


if __name__ == "__main__":
    main()
