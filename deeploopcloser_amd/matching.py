"""encode() / match() / match_topk(): the surface BASELINE.json's north_star
names.  The reference has no symbol with these names; they are
defined here as
    encode(frames)            := SDAV.transform / CnnVtl.transform
    match(desc_q, desc_db)    := dense cosine-similarity matrix
    match_topk(q, db, k)      := top-k cosine match (scores desc, ties -> lower index)
and a resident KeyframeDatabase for repeated queries against one shard.
"""
import numpy as np
import torch

from .engine import default_engine, torch_dtype


def encode(frames, network):
    """Descriptors of `frames` with a SDAV / DA / CnnVtl instance (its transform())."""
    return network.transform(frames)


def flatten_frame_descriptors(h, patches=30):
    """SDAV.transform returns the FLAT [B*30, H] array (SDAV.py:163); one
    place descriptor per frame is its [30*H] concatenation."""
    h = h if isinstance(h, torch.Tensor) else torch.from_numpy(np.asarray(h))
    return h.reshape(h.shape[0] // patches, patches * h.shape[1])


class KeyframeDatabase:
    """A shard of stored (L2-normalised bf16 / fp16) key-frame descriptors resident in HBM."""

    def __init__(self, descriptors, dtype="bf16", center=False, row_offset=0, device=None, stored=False):
        self.engine = default_engine(device)
        self.dtype = torch_dtype(dtype)
        self.center = center
        self.row_offset = int(row_offset)
        if stored:
            d = descriptors.to(self.engine.device)
            if d.dtype != self.dtype:
                raise ValueError("stored descriptors have dtype %s, expected %s" % (d.dtype, self.dtype))
            self.rows = d
        else:
            x = self.engine.to_device(descriptors)
            if x.dtype not in (torch.float32, torch.float64):
                x = x.to(torch.float32)
            self.rows = self.engine.normalize(x, self.dtype, center)

    def __len__(self):
        return self.rows.shape[0]

    def prepare_queries(self, queries):
        x = self.engine.to_device(queries)
        if x.dtype in (torch.bfloat16, torch.float16):
            return x
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        return self.engine.normalize(x, self.dtype, self.center)

    def match_topk(self, queries, k, out=None):
        q = self.prepare_queries(queries)
        return self.engine.match_topk(q, self.rows, k, self.row_offset, out=out)

    def match(self, queries):
        return self.engine.cosine_scores(self.prepare_queries(queries), self.rows)


def match(desc_q, desc_db, dtype="bf16", center=False):
    """Dense cosine-similarity matrix [Q, N] (float32 numpy)."""
    db = KeyframeDatabase(desc_db, dtype=dtype, center=center)
    return db.match(desc_q).cpu().numpy()


def match_topk(desc_q, desc_db, k, dtype="bf16", center=False):
    """(scores [Q,k] float32, idx [Q,k] int64) numpy arrays."""
    db = KeyframeDatabase(desc_db, dtype=dtype, center=center)
    s, i = db.match_topk(desc_q, k)
    return s.cpu().numpy(), i.cpu().numpy()
