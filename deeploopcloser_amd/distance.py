"""cnn_vtl descriptor distance with the reference's call surface
(src/cnn_vtl/similarity/DistanceCalculator.py:8-12) on MI355X, plus the full
N x N loop of src/cnn_vtl/create_distance_matrix.py:30-36 as one call."""
import numpy as np
import torch

from .engine import default_engine


class _PairBuffers:
    """Per-length buffers of the per-pair entry point: a caller that keeps the reference's O(N^2) Python loop
    (create_distance_matrix.py:30-36) calls it once per pair, so nothing is allocated per call -- a page-locked host row
    pair, its device image (rows padded to 16 bytes), the 2 x 2 result and a page-locked word to read it back into."""

    def __init__(self, engine, n):
        self.n = n
        self.ld = (n + 15) // 16 * 16
        self.host = torch.zeros((2, self.ld), dtype=torch.int8).pin_memory()
        self.host_np = self.host.numpy()
        self.dev = torch.zeros((2, self.ld), dtype=torch.int8, device=engine.device)
        self.out = torch.empty((2, 2), dtype=torch.int64, device=engine.device)
        self.res = torch.zeros((1,), dtype=torch.int64).pin_memory()


class DistanceCalculator:
    _pair = {}

    @staticmethod
    def calculate_distance(desc1, desc2):
        """sum_k popcount(|a_k ^ b_k|) on int8 (DistanceCalculator.py:4-12) -> numpy int64."""
        a = np.asarray(desc1, dtype=np.int8).reshape(-1)
        b = np.asarray(desc2, dtype=np.int8).reshape(-1)
        n = min(a.size, b.size)                                   # zip() stops at the shorter one
        if n == 0:
            return np.int64(0)
        e = default_engine()
        key = (e.device.index, n)
        buf = DistanceCalculator._pair.get(key)
        if buf is None:
            if len(DistanceCalculator._pair) > 16:
                DistanceCalculator._pair.clear()
            buf = DistanceCalculator._pair[key] = _PairBuffers(e, n)
        buf.host_np[0, :n] = a[:n]
        buf.host_np[1, :n] = b[:n]
        buf.dev.copy_(buf.host, non_blocking=True)
        e.cnnvtl_distance_matrix(buf.dev, d=n, out=buf.out)
        buf.res.copy_(buf.out.view(-1)[1:2], non_blocking=True)
        torch.cuda.current_stream(e.device).synchronize()
        return np.int64(buf.res[0].item())

    @staticmethod
    def distance_matrix(descriptors):
        """Full N x N matrix incl. the diagonal (create_distance_matrix.py:30-36), int64."""
        e = default_engine()
        d = e.to_device(descriptors, torch.int8)
        if d.dim() != 2:
            raise ValueError("descriptors must be [N, D] int8")
        return e.cnnvtl_distance_matrix(d).cpu().numpy()
