"""SDAV similarity with the reference's call surface
(src/sdav/similarity/SimilarityCalculator.py:4-17) on MI355X, plus the
all-vs-all matrix of src/sdav/create_similarity_matrix.py:29-38 as one call."""
import numpy as np
import torch

from .engine import default_engine


class SimilarityCalculator:
    def __init__(self, dataset: np.ndarray, mu=0.5, sigma=0.2, a=10, b=-10, device=None):
        self.mu, self.sigma, self.a, self.b = mu, sigma, a, b
        self.engine = default_engine(device)
        self._pair = None
        self.dataset = dataset

    @property
    def dataset(self):
        return self._dataset

    @dataset.setter
    def dataset(self, dataset):
        """The reference reads self.dataset (and mu / sigma) on EVERY similarity_score call (:13-14); here the average
        response and the distinctive score are hoisted out of the pair loop, so assigning the public attribute
        re-hoists them: `calc.dataset = other` scores against `other` from the next call on, as upstream."""
        ds = self.engine.to_device(dataset, torch.float64)
        if ds.dim() != 3:
            raise ValueError("dataset must be [N, P, H]")
        self._dataset, self._dataset_dev = dataset, ds
        self._pair = None                                     # the pair buffers are sized by the dataset's width
        # ... and what that pass learned about the dataset's range serves similarity_matrix() over the same tensor
        self._score, self._range = self.engine.distinctive_score(ds, self.mu, self.sigma, with_range=True)
        self._hoisted = (self.mu, self.sigma)

    def _current_score(self):
        if self._hoisted != (self.mu, self.sigma):            # mu / sigma are public attributes too (:25-27)
            self._score, self._range = self.engine.distinctive_score(self._dataset_dev, self.mu, self.sigma, with_range=True)
            self._hoisted = (self.mu, self.sigma)
        return self._score

    def similarity_score(self, h1, h2):
        """similarity_score(h1, h2) (:12-17): python float, +inf when a matched pair is identical.
        The per-pair entry of a caller that keeps the reference's loop (create_similarity_matrix.py:34-38): the frame
        pair goes through page-locked buffers kept with the calculator, the 2 x 2 matrix call never reads a flag back
        (DLC_SIM_NO_HOST_SYNC) and one 8-byte copy brings the score home."""
        p, h = self._dataset_dev.shape[1], self._dataset_dev.shape[2]
        as_np = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
        a, b = np.asarray(as_np(h1), dtype=np.float64), np.asarray(as_np(h2), dtype=np.float64)
        if a.shape != b.shape or a.ndim != 2 or a.shape[1] != h:
            if a.ndim == 2 and a.shape[1] != h:
                raise ValueError("descriptor width differs from the dataset's")
            raise ValueError("h1 and h2 must both be [P, %d]" % h)
        buf = self._pair
        if buf is None or buf["host"].shape[1] != a.shape[0]:
            buf = self._pair = {"host": torch.empty((2, a.shape[0], h), dtype=torch.float64).pin_memory(),
                                "dev": torch.empty((2, a.shape[0], h), dtype=torch.float64, device=self.engine.device),
                                "stats": torch.zeros((2,), dtype=torch.int64, device=self.engine.device),
                                "res": torch.zeros((3,), dtype=torch.float64).pin_memory()}
            buf["host_np"] = buf["host"].numpy()
        buf["host_np"][0] = a
        buf["host_np"][1] = b
        buf["dev"].copy_(buf["host"], non_blocking=True)
        score = self._current_score()
        out, _ = self.engine.sdav_similarity_matrix(buf["dev"], score, self.a, self.b, want_int64=False,
                                                    no_host_sync=True, stats=buf["stats"])
        buf["res"][:1].copy_(out.view(-1)[1:2], non_blocking=True)
        buf["res"][1:].copy_(buf["stats"].to(torch.float64), non_blocking=True)
        torch.cuda.current_stream(self.engine.device).synchronize()
        if buf["res"][2].item() != 0.0:            # a NaN / infinity in the pair: the fp64 form reproduces NumPy's handling
            out, _ = self.engine.sdav_similarity_matrix(buf["dev"], score, self.a, self.b, want_int64=False, force_f64=True)
            return float(out[0, 1].item())
        return float(buf["res"][0].item())

    def similarity_matrix(self, descriptors=None, as_int64=True):
        """create_similarity_matrix.py:29-38: scores for i<j mirrored, diagonal -1.
        as_int64=True returns the reference's int64 matrix (truncated scores)."""
        d = self._dataset_dev if descriptors is None else self.engine.to_device(descriptors, torch.float64)
        f, i = self.engine.sdav_similarity_matrix(d, self._current_score(), self.a, self.b, want_int64=as_int64,
                                                  range=self._range if descriptors is None else None)
        return (i if as_int64 else f).cpu().numpy()


class SimilarityStream:
    """The reference's similarity for frames that ARRIVE ONE BY ONE (a robot adding key-frames): each new frame is scored
    against every older one -- row[j] = SimilarityCalculator.similarity_score(h_j, h_new) (SimilarityCalculator.py:12-49),
    the entries the loop of create_similarity_matrix.py:34-38 would add to its matrix for that frame, bit for bit -- without
    touching the older frames again: their descriptors and the filter's fixed-point panel stay resident in HBM and are
    appended to (dlc_sdav_stream_* in include/dlc.h).

    score_source: the dataset [N, P, H] whose distinctive score weights the distances (what SimilarityCalculator(dataset)
    computes, :20-27), or that score vector [H] itself; it is fixed for the life of the stream.  value_range: the range the
    descriptors live in (SDAV outputs are sigmoid values: (0, 1)); a value outside it poisons the stream (queries return
    NaN and say so in .stats) -- the filter's error bound is stated for that range.  column_centre ([H], optional): the
    range then bounds x - column_centre[k] instead of x: low-contrast descriptors (every column close to its own mean)
    want their column means here and a narrow value_range around 0, since the filter's error window is a fixed fraction of
    the range's width squared."""

    def __init__(self, score_source, patches=30, width=2500, capacity=1024, value_range=(0.0, 1.0), mu=0.5, sigma=0.2, a=10,
                 b=-10, device=None, column_centre=None):
        self.engine = default_engine(device)
        eng = self.engine
        self.a, self.b = a, b
        self.p, self.h = int(patches), int(width)
        self.range = (float(value_range[0]), float(value_range[1]))
        self.centre = None if column_centre is None else eng.to_device(column_centre, torch.float64).contiguous()
        if self.centre is not None and tuple(self.centre.shape) != (self.h,):
            raise ValueError("column_centre must have the descriptor width %d" % self.h)
        src = eng.to_device(score_source, torch.float64)
        if src.dim() == 1:
            if src.shape[0] != self.h:
                raise ValueError("score vector must have the descriptor width %d" % self.h)
            self.score = src.contiguous()
        elif src.dim() == 3 and src.shape[2] == self.h:
            self.score = eng.distinctive_score(src, mu, sigma)
        else:
            raise ValueError("score_source must be a dataset [N, P, %d] or a score vector [%d]" % (self.h, self.h))
        self.stats = torch.zeros((2,), dtype=torch.int64, device=eng.device)
        self._n = 0
        self._alloc(int(capacity))

    def _alloc(self, capacity):
        eng = self.engine
        desc = torch.zeros((capacity, self.p, self.h), dtype=torch.float64, device=eng.device)
        state = eng.sdav_stream_state(capacity, self.p, self.h, *self.range, col_centre=self.centre)
        if self._n:
            desc[:self._n] = self.desc[:self._n]
            eng.sdav_stream_append(state, desc, 0, self._n, self.score)      # growing re-quantises once (amortised)
        self.desc, self.state, self.capacity = desc, state, capacity

    def __len__(self):
        return self._n

    def append(self, frames):
        """Frames [B, P, H] (or one [P, H]) become resident; returns the index of the first."""
        x = self.engine.to_device(frames, torch.float64)
        if x.dim() == 2:
            x = x.unsqueeze(0)
        if x.dim() != 3 or x.shape[1] != self.p or x.shape[2] != self.h:
            raise ValueError("frames must be [B, %d, %d]" % (self.p, self.h))
        first, b = self._n, x.shape[0]
        if first + b > self.capacity:
            self._alloc(max(2 * self.capacity, first + b))
        self.desc[first:first + b] = x
        self.engine.sdav_stream_append(self.state, self.desc, first, first + b, self.score)
        self._n = first + b
        return first

    def query(self, f=None, out=None):
        """Device tensor [f] of score(h_j, h_f), j < f, for the resident frame f (default: the newest); out: a contiguous
        fp64 device tensor of f entries to write into."""
        f = self._n - 1 if f is None else int(f)
        if not 0 <= f < self._n:
            raise ValueError("frame %d is not resident (0..%d)" % (f, self._n - 1))
        if out is not None and (out.dtype != torch.float64 or out.numel() != f or not out.is_contiguous()):
            raise ValueError("query: out must be a contiguous float64 tensor of %d entries" % f)
        return self.engine.sdav_stream_query(self.state, self.desc, f, self.score, self.a, self.b, out=out, stats=self.stats)

    def query_batch(self, first, count):
        """Device tensor [count, first + count - 1]: row q = query(first + q) in its first `first + q` entries (the rest of a
        row is unspecified) -- the resident frames first .. first + count - 1 scored in ONE pair of launches."""
        first, count = int(first), int(count)
        if count < 1 or first < 0 or first + count > self._n:
            raise ValueError("frames %d .. %d are not all resident (0..%d)" % (first, first + count - 1, self._n - 1))
        return self.engine.sdav_stream_query_batch(self.state, self.desc, first, count, self.score, self.a, self.b, stats=self.stats)

    def query_and_insert(self, frame):
        """One new frame [P, H]: it becomes resident and its row against all older frames comes back (device, fp64)."""
        self.append(frame)
        return self.query()

    def similarity_row(self, frame):
        """Alias of query_and_insert, as a NumPy row (the reference's scores are host floats)."""
        return self.engine.download(self.query_and_insert(frame)) if self._n > 0 else np.empty(0)
