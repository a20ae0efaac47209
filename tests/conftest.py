import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def topk_flips_are_ties(got_idx, want_idx, full_scores, want_scores, tol, row_offset=0):
    """Long descriptors (the reference's 75 000-d place descriptors) make near-ties that fp32 accumulation
    cannot resolve: the fp32 sum of 75 008 products is good to ~5e-6, and two key-frames whose exact fp64 scores
    are closer than that may change places.  Returns (number of differing slots, True if at every differing
    slot the row the GPU put there has an exact score within `tol` of the oracle's score for that slot)."""
    got_idx, want_idx = np.asarray(got_idx), np.asarray(want_idx)
    diff = got_idx != want_idx
    if not diff.any():
        return 0, True
    rows, cols = np.nonzero(diff)
    got_true = full_scores[rows, got_idx[rows, cols] - row_offset]
    return int(diff.sum()), bool(np.abs(got_true - want_scores[rows, cols]).max() < tol)
