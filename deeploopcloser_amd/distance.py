"""cnn_vtl descriptor distance with the reference's call surface
(src/cnn_vtl/similarity/DistanceCalculator.py:8-12) on MI355X, plus the full
N x N loop of src/cnn_vtl/create_distance_matrix.py:30-36 as one call."""
import numpy as np
import torch

from .engine import default_engine


class DistanceCalculator:
    @staticmethod
    def calculate_distance(desc1, desc2):
        """sum_k popcount(|a_k ^ b_k|) on int8 (DistanceCalculator.py:4-12) -> numpy int64."""
        a = np.asarray(desc1, dtype=np.int8).reshape(-1)
        b = np.asarray(desc2, dtype=np.int8).reshape(-1)
        n = min(a.size, b.size)                                   # zip() stops at the shorter one
        if n == 0:
            return np.int64(0)
        e = default_engine()
        m = e.cnnvtl_distance_matrix(torch.from_numpy(np.stack([a[:n], b[:n]])).to(e.device))
        return np.int64(m[0, 1].item())

    @staticmethod
    def distance_matrix(descriptors):
        """Full N x N matrix incl. the diagonal (create_distance_matrix.py:30-36), int64."""
        e = default_engine()
        d = e.to_device(descriptors, torch.int8)
        if d.dim() != 2:
            raise ValueError("descriptors must be [N, D] int8")
        return e.cnnvtl_distance_matrix(d).cpu().numpy()
