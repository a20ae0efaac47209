"""SDAV similarity with the reference's call surface
(src/sdav/similarity/SimilarityCalculator.py:4-17) on MI355X, plus the
all-vs-all matrix of src/sdav/create_similarity_matrix.py:29-38 as one call."""
import numpy as np
import torch

from .engine import default_engine


class SimilarityCalculator:
    def __init__(self, dataset: np.ndarray, mu=0.5, sigma=0.2, a=10, b=-10, device=None):
        self.mu, self.sigma, self.a, self.b = mu, sigma, a, b
        self.dataset = dataset
        self.engine = default_engine(device)
        ds = self.engine.to_device(dataset, torch.float64)
        if ds.dim() != 3:
            raise ValueError("dataset must be [N, P, H]")
        self._dataset_dev = ds
        # hoisted: the reference recomputes these for every pair (:13-14)
        self._score = self.engine.distinctive_score(ds, mu, sigma)

    def similarity_score(self, h1, h2):
        """similarity_score(h1, h2) (:12-17): python float, +inf when a matched pair is identical."""
        pair = torch.stack([self.engine.to_device(h1, torch.float64), self.engine.to_device(h2, torch.float64)])
        if pair.shape[2] != self._dataset_dev.shape[2]:
            raise ValueError("descriptor width differs from the dataset's")
        out, _ = self.engine.sdav_similarity_matrix(pair, self._score, self.a, self.b, want_int64=False)
        return float(out[0, 1].item())

    def similarity_matrix(self, descriptors=None, as_int64=True):
        """create_similarity_matrix.py:29-38: scores for i<j mirrored, diagonal -1.
        as_int64=True returns the reference's int64 matrix (truncated scores)."""
        d = self._dataset_dev if descriptors is None else self.engine.to_device(descriptors, torch.float64)
        f, i = self.engine.sdav_similarity_matrix(d, self._score, self.a, self.b, want_int64=as_int64)
        return (i if as_int64 else f).cpu().numpy()
