"""Streaming loop-closure queries over the resident top-k engine.

The reference stops at the all-vs-all matrices (create_similarity_matrix.py:29-38,
create_distance_matrix.py:30-36); what a SLAM front-end asks is the streaming form of the same
comparison: "does the frame that just arrived look like a place seen a while ago?".  A
LoopClosureDetector keeps every key-frame's descriptor resident in HBM (KeyframeDatabase.append),
matches each new frame against the key-frames more than `exclusion` frames older with the MFMA
top-k path and reports the candidates whose cosine score reaches `threshold`.

    python -m deeploopcloser_amd.loop_closure DATASET_DIR --network cnn_vtl --k 5 --threshold 0.9

Batching never changes a result: a batch of B frames is matched in one call against the longest
prefix any of its frames may see, each frame keeping to the key-frames old enough for IT inside the
selection (dlc_cosine_topk_older).  The same lists come from a match with k+B-1 candidates per frame
followed by each frame's first k candidates that are old enough (first_k_eligible below, the form
the engine call is tested against: at most B-1 of the extra rows are too recent for a frame).
"""
import argparse
import contextlib
import glob
import os
import sys
import time

import numpy as np
import torch

from . import _lib as L
from .matching import KeyframeDatabase


def first_k_eligible(scores, idx, limit, k):
    """Rows of (scores, idx) [B, kk] sorted best-first -> the first k entries per row whose id is
    below that row's limit [B], order kept; missing slots are (-inf, -1)."""
    ok = (idx >= 0) & (idx < limit.unsqueeze(1))
    order = torch.argsort((~ok).to(torch.int8), dim=1, stable=True)[:, :k]
    s, i, ok = scores.gather(1, order), idx.gather(1, order), ok.gather(1, order)
    s = torch.where(ok, s, torch.full_like(s, float("-inf")))
    i = torch.where(ok, i, torch.full_like(i, -1))
    if s.shape[1] < k:                                   # fewer candidates than k were requested
        pad = k - s.shape[1]
        s = torch.cat([s, s.new_full((s.shape[0], pad), float("-inf"))], 1)
        i = torch.cat([i, i.new_full((i.shape[0], pad), -1)], 1)
    return s, i


class LoopClosureDetector:
    def __init__(self, dim, k=5, threshold=0.9, exclusion=30, dtype="bf16", center=False, capacity=4096,
                 device=None):
        if not 1 <= k <= L.DLC_MAX_K:
            raise ValueError("k=%d outside 1..%d" % (k, L.DLC_MAX_K))
        if exclusion < 0:
            raise ValueError("exclusion must be >= 0")
        self.k, self.threshold, self.exclusion = int(k), float(threshold), int(exclusion)
        self.db = KeyframeDatabase.empty(dim, capacity=capacity, dtype=dtype, center=center, device=device)

    def __len__(self):
        return len(self.db)

    @property
    def max_batch(self):
        """Largest number of frames one engine call takes (one 256-query tile of the score pass)."""
        return 256

    def query_and_insert(self, descriptors):
        """The next B frames' descriptors [B, dim] (ids len(self) .. len(self)+B-1) ->
        (scores [B,k] float32, ids [B,k] int64) on the device, best first, (-inf, -1) where fewer
        than k key-frames are old enough; the frames are then key-frames themselves."""
        x = self.db._as_float(descriptors)
        if x.dim() != 2:
            raise ValueError("descriptors must be [B, dim]")
        out_s, out_i = [], []
        for lo in range(0, x.shape[0], self.max_batch):
            s, i = self._step(x[lo:lo + self.max_batch])
            out_s.append(s)
            out_i.append(i)
        if not out_s:
            dev = self.db.engine.device
            return (torch.empty((0, self.k), dtype=torch.float32, device=dev),
                    torch.empty((0, self.k), dtype=torch.int64, device=dev))
        if len(out_s) == 1:                                  # (torch.cat of one tensor is a copy: two launches per batch)
            return out_s[0], out_i[0]
        return torch.cat(out_s), torch.cat(out_i)

    def _step(self, x):
        db, k = self.db, self.k
        b = x.shape[0]
        g0, _ = db.append(x)                               # normalised once, used as query and as key-frame
        g0 -= db.row_offset
        q = db.rows[g0:g0 + b]
        n_search = g0 + b - 1 - self.exclusion             # what the newest frame of the batch may see
        dev = db.engine.device
        if n_search <= 0:
            return (torch.full((b, k), float("-inf"), dtype=torch.float32, device=dev),
                    torch.full((b, k), -1, dtype=torch.int64, device=dev))
        # one score pass over what the newest frame may see; frame j of the batch keeps to the rows below g0 - exclusion + j
        # (dlc_cosine_topk_older -- the lists of a k + b - 1 match followed by dlc_topk_keep_older / first_k_eligible)
        return db.engine.match_topk(q, db.rows[:n_search], k, older_than=g0 - self.exclusion)

    def loops(self, scores, ids, first_id):
        """[(frame id, matched key-frame id, score)] of the candidates at or above the threshold."""
        s, i = scores.cpu().numpy(), ids.cpu().numpy()
        out = []
        for r in range(s.shape[0]):
            for c in range(s.shape[1]):
                if i[r, c] >= 0 and s[r, c] >= self.threshold:
                    out.append((first_id + r, int(i[r, c]), float(s[r, c])))
        return out


class SdavLoopClosureDetector:
    """The same question asked with the REFERENCE's similarity (SimilarityCalculator.similarity_score,
    src/sdav/similarity/SimilarityCalculator.py:12-49) instead of the cosine of flattened descriptors: every new frame's
    [P, H] SDAV descriptors are scored against all resident frames more than `exclusion` frames older through the
    streaming filter (similarity.SimilarityStream: the older frames' panel is resident, nothing is re-quantised), and the
    k best (score descending, ties -> the older frame) at or above `threshold` are the loop candidates."""

    def __init__(self, score_source, patches=30, width=2500, k=5, threshold=float("-inf"), exclusion=30, capacity=1024,
                 device=None, **stream_args):
        from .similarity import SimilarityStream
        if k < 1:
            raise ValueError("k must be >= 1")
        if exclusion < 0:
            raise ValueError("exclusion must be >= 0")
        self.k, self.threshold, self.exclusion = int(k), float(threshold), int(exclusion)
        self.stream = SimilarityStream(score_source, patches=patches, width=width, capacity=capacity, device=device,
                                       **stream_args)

    def __len__(self):
        return len(self.stream)

    def query_and_insert(self, frames):
        """frames [B, P, H] (ids len(self) .. + B - 1) -> (scores [B, k] float64, ids [B, k] int64) on the device, best first,
        (-inf, -1) where fewer than k frames are old enough; the frames are resident afterwards.  A POISONED stream (a value
        outside its fixed range or a NaN was appended, now or earlier: `poisoned`, SimilarityStream.stats[1]) returns
        (NaN, -1) in every slot -- "these scores mean nothing", visible in the tensors themselves without a host read;
        loops() raises."""
        st = self.stream
        eng = st.engine
        x = eng.to_device(frames, torch.float64)
        if x.dim() == 2:
            x = x.unsqueeze(0)
        b = x.shape[0]
        first = st.append(x)                                          # all B frames become resident: one quantisation launch
        if first + b - 1 == 0:                                        # the very first frame alone: nothing older
            none = torch.full((b, self.k), float("-inf"), dtype=torch.float64, device=eng.device)
            return (torch.where(self.poisoned != 0, float("nan"), none),
                    torch.full((b, self.k), -1, dtype=torch.int64, device=eng.device))
        # frame first + r against every older frame, all B of them in one pair of launches (dlc_sdav_stream_query_batch):
        # rows[r, :first + r]
        rows = st.query_batch(first, b)
        # the k best of the frames old enough -- one launch for the batch (dlc_topk_rows_f64: score descending, ties ->
        # the older frame; the kernel reads the stream's poison word and answers (NaN, -1) everywhere when it is set)
        return eng.topk_rows_f64(rows, first - self.exclusion, 1, self.k, poison=self.poisoned)

    # ---- two batches in flight ------------------------------------------------------------------------------------------
    # query_and_insert runs a batch's six launches one behind the other: copy, quantisation, the strip's product kernel (200
    # of the 275 us at 32 frames against 1063), resolution, scores, ranking.  Only the product kernel needs the whole chip.
    # submit() / result() give it the engine's second stream to itself and keep the rest on the caller's: batch b's copy +
    # quantisation run beside batch b - 1's products, batch b - 1's resolution + scores + ranking beside batch b's (what
    # may run beside what: include/dlc.h, dlc_sdav_stream_query_batch_staged).  Everything the caller touches -- the frames
    # going in, the lists coming out -- lives on the caller's stream; the second stream only ever carries product kernels.
    # The lists are the lists query_and_insert returns, bit for bit.
    PIPELINE_MIN_BATCH = 8                                            # below it there is no strip (include/dlc.h)

    def submit(self, frames):
        """frames [B, P, H] -> ticket.  The frames become resident; result(ticket) hands out (scores [B, k], ids [B, k]) --
        to be fetched before the second submit() after this one (a slot's buffers are reused then).  Batches of fewer than
        8 frames, the very first batch and a batch that makes the stream grow go through query_and_insert (no overlap)."""
        st, eng = self.stream, self.stream.engine
        x = eng.to_device(frames, torch.float64)
        if x.dim() == 2:
            x = x.unsqueeze(0)
        if x.dim() != 3 or x.shape[1] != st.p or x.shape[2] != st.h:       # (before a ticket is spent on it)
            raise ValueError("frames must be [B, %d, %d]" % (st.p, st.h))
        b, first = x.shape[0], len(st)
        if not hasattr(self, "_slots"):
            self._slots, self._pending, self._tickets = [{}, {}], None, 0
        t = self._tickets
        self._tickets += 1
        main, side = torch.cuda.current_stream(eng.device), eng.side_stream
        slot = self._slots[t % 2]
        if b < self.PIPELINE_MIN_BATCH or first == 0 or first + b > st.capacity:
            self._flush()                                              # (a growing stream re-quantises everything: nothing in flight)
            main.wait_stream(side)
            slot.update({"ticket": t, "out": self.query_and_insert(x)})
            return t
        need = eng.lib.dlc_sdav_stream_query_batch_workspace_bytes(st.capacity, st.p, b)
        if slot.get("ws") is None or slot["ws"].numel() < need:
            slot["ws"] = torch.empty(int(need), dtype=torch.uint8, device=eng.device)
        if slot.get("rows") is None or slot["rows"].shape[0] < b or slot["rows"].shape[1] < st.capacity:
            slot["rows"] = torch.empty((b, st.capacity), dtype=torch.float64, device=eng.device)
        st.append(x)                                                   # copy + quantisation: beside the previous batch's products
        quantised = torch.cuda.Event()
        quantised.record(main)                                         # (also behind the last use of this slot's buffers, two tickets ago)
        side.wait_event(quantised)
        rows = slot["rows"][:b]
        eng.sdav_stream_query_batch_staged(st.state, st.desc, first, b, st.score, 1, rows, slot["ws"], st.a, st.b, stats=st.stats,
                                           stream=side)
        products = torch.cuda.Event()
        products.record(side)
        self._flush()                                                  # the previous batch's second half: beside this batch's products
        slot.update({"ticket": t, "first": first, "b": b, "products": products, "out": None})
        self._pending = t
        return t

    def _flush(self):
        """The second half of the batch whose products are in flight -- resolution + scores + ranking, on the caller's stream
        behind that batch's product kernel."""
        if getattr(self, "_pending", None) is None:
            return
        st, eng = self.stream, self.stream.engine
        slot = self._slots[self._pending % 2]
        torch.cuda.current_stream(eng.device).wait_event(slot["products"])
        rows = slot["rows"][:slot["b"]]
        eng.sdav_stream_query_batch_staged(st.state, st.desc, slot["first"], slot["b"], st.score, 2, rows, slot["ws"], st.a, st.b,
                                           stats=st.stats)
        slot["out"] = eng.topk_rows_f64(rows, slot["first"] - self.exclusion, 1, self.k, poison=self.poisoned)
        self._pending = None

    def result(self, ticket):
        """(scores [B, k] float64, ids [B, k] int64) of a submitted batch, in the current stream's order."""
        if not hasattr(self, "_slots") or not self._tickets - 2 <= ticket < self._tickets:
            raise ValueError("SdavLoopClosureDetector.result: ticket %r is not in flight" % (ticket,))
        if self._pending == ticket:
            self._flush()
        slot = self._slots[ticket % 2]
        if slot.get("ticket") != ticket or slot.get("out") is None:
            raise ValueError("SdavLoopClosureDetector.result: ticket %r is not in flight" % (ticket,))
        return slot["out"]

    @property
    def poisoned(self):
        """Device int64 [1] (a view of SimilarityStream.stats): non-zero once a descriptor value outside the stream's fixed
        range (or a NaN / infinity) has been appended.  No host synchronisation to look at it on the device."""
        return self.stream.stats[1:2]

    def _check_poison(self):
        if int(self.stream.stats[1]) != 0:
            raise RuntimeError("SdavLoopClosureDetector: a descriptor value outside the stream's fixed range (or a NaN / "
                               "infinity) was appended -- the filter's error bound does not hold, every later row is NaN; "
                               "create the stream with a value_range / column_centre that covers the data")

    def loops(self, scores, ids, first_id):
        """[(frame id, older frame id, score)] at or above the threshold; raises when the stream has been poisoned (a value
        outside its fixed range: SimilarityStream.stats[1])."""
        s, i = scores.cpu().numpy(), ids.cpu().numpy()
        self._check_poison()
        return [(first_id + r, int(i[r, c]), float(s[r, c])) for r in range(s.shape[0]) for c in range(s.shape[1])
                if i[r, c] >= 0 and s[r, c] >= self.threshold]


def _frame_files(dataset_path, pattern):
    files = sorted(glob.glob(os.path.join(dataset_path, pattern)))
    if not files:
        raise ValueError("Specified dataset is empty or could not find dataset")        # InputGenerator.py:21-23
    return files


def describe_sdav(files, network=None, key_points_fn=None):
    """Frames -> one [30*2500] place descriptor per frame (patches -> SDAV.transform, flattened), a DEVICE tensor: the
    uint8 frames go up once, grey / key-points / patches / encoder run back to back in HBM (pipeline.py)."""
    from . import pipeline
    from .input import read_ppm
    from .sdav import SDAV
    network = network or SDAV()
    p = network.input_shape[0]
    frames = [read_ppm(f) for f in files]
    if any(fr.shape != frames[0].shape for fr in frames):
        # a chunk of frames of several sizes (the reference parses them one by one, CvInputParser.py:30-33): each frame's
        # patches are gathered on the device on its own, as drivers.create_similarity_matrix does
        from .input import CvInputParser
        parser = CvInputParser(p, int(round(np.sqrt(network.input_shape[1]))))
        x = torch.stack([parser.parse_tensor(fr, key_points_fn(fr.shape[:2]) if key_points_fn else None) for fr in frames])
        return network.transform_tensor(x).view(len(files), -1)
    kp = None
    if key_points_fn:
        kp = pipeline.key_point_array([key_points_fn(fr.shape[:2]) for fr in frames], p, network.engine)
    desc = pipeline.sdav_descriptors_from_frames(np.stack(frames), network, key_points=kp)
    return desc.view(len(files), -1)


def describe_cnn_vtl(files, network=None):
    """Frames -> CnnVtl int8 descriptors [B, D'] as float32 for the cosine engine, a DEVICE tensor (pipeline.py)."""
    from . import pipeline
    from .cnn_vtl import CnnVtl
    from .input import read_ppm
    frames = np.stack([read_ppm(f)[..., ::-1] for f in files])           # BGR, as create_distance_matrix.py:23
    network = network or CnnVtl(input_shape=[len(files)] + list(frames.shape[1:]))
    return pipeline.cnn_vtl_descriptors_from_frames(frames, network).to(torch.float32)


def main(argv=None):
    ap = argparse.ArgumentParser(description="stream the frames of a dataset through the loop-closure detector")
    ap.add_argument("dataset_path")
    ap.add_argument("--pattern", default="*.ppm")
    ap.add_argument("--network", choices=["sdav", "cnn_vtl"], default="cnn_vtl")
    ap.add_argument("--weights", help=".npz written by SDAV.save_weights (sdav) / AlexNet .npy blob (cnn_vtl)")
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--threshold", type=float, default=0.9)
    ap.add_argument("--exclusion", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16, help="frames encoded and matched per step")
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16")
    ap.add_argument("--no-latency-mode", action="store_true",
                    help="keep the one-pass GEMMs (bit-identical to a large-batch encode) for small batches too")
    args = ap.parse_args(argv)

    files = _frame_files(args.dataset_path, args.pattern)
    from .engine import default_engine
    eng = default_engine()
    latency = args.batch <= 16 and not args.no_latency_mode      # split-K encode GEMMs: ~8x lower latency per frame
    with (eng.latency_mode() if latency else contextlib.nullcontext()):
        return _stream(args, files)


def _stream(args, files):
    if args.network == "sdav":
        from .sdav import SDAV
        net = SDAV()
        if args.weights:
            net.load_weights(args.weights)
        describe = lambda fs: describe_sdav(fs, net)
    else:
        from .cnn_vtl import CnnVtl
        from .input import read_ppm
        shape = read_ppm(files[0]).shape
        net = CnnVtl(input_shape=[args.batch] + list(shape))
        if args.weights:
            net.load_alexnet_npy(args.weights)
        describe = lambda fs: describe_cnn_vtl(fs, net)
    det = None
    lat = []                                                     # wall-clock per step: file read -> descriptors -> match -> candidates on the host
    for lo in range(0, len(files), args.batch):
        chunk = files[lo:lo + args.batch]
        t0 = time.perf_counter()
        desc = describe(chunk)
        if det is None:
            det = LoopClosureDetector(desc.shape[1], k=args.k, threshold=args.threshold, exclusion=args.exclusion,
                                      dtype=args.dtype, center=True, capacity=max(4096, len(files)))
        s, i = det.query_and_insert(desc)
        found = det.loops(s, i, lo)                              # (the one host read of the step)
        lat.append(((time.perf_counter() - t0) * 1e3, len(chunk)))
        for frame, match, score in found:
            print("loop\t%d\t%s\t%d\t%s\t%.4f" % (frame, os.path.basename(files[frame]), match,
                                                 os.path.basename(files[match]), score))
    print("frames\t%d\tkey-frames\t%d" % (len(files), len(det)), file=sys.stderr)
    # the first step pays one-time costs (the library's first launches, workspaces, the database's reservation): reported apart
    steady = lat[1:] if len(lat) > 1 else lat
    ms = np.array([m for m, _ in steady])
    per_frame = float(ms.sum() / max(1, sum(c for _, c in steady)))
    print("latency\tbatch\t%d\tsteps\t%d\tfirst_step_ms\t%.2f\tms_per_step_median\t%.3f\tms_per_step_max\t%.3f\tms_per_frame\t%.3f"
          % (args.batch, len(lat), lat[0][0], float(np.median(ms)), float(ms.max()), per_frame), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
