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
    """A shard of stored (L2-normalised bf16 / fp16) key-frame descriptors resident in HBM.

    The rows live in a capacity-reserved buffer so that a running loop-closure session can
    append() key-frames without re-uploading the shard: `rows` is the view of the first
    len(db) rows, growth doubles the reservation (sized against 288 GB of HBM, a 4096-d bf16
    key-frame is 8 KiB).

    Norms.  The top-k's certificate (include/dlc.h, NORMS) is derived for rows of norm <= 1.005 -- what the engine's own
    normaliser writes.  Rows handed over with stored=True (a loaded shard, descriptors stored by another tool, a slice of
    anything) are not taken on trust: one reduction at construction measures their largest norm (`norm_bound`, a device
    float, never read by the host) and every match of this database then certifies with tau scaled by |q| * norm_bound
    (dlc_cosine_tau_scale) -- the result is the exact fp64 top-k whatever the rows' norms.  The same goes for queries that
    arrive already stored (bf16 / fp16) from anywhere but normalize().  norm_bound is None when every row is the normaliser's."""

    def __init__(self, descriptors, dtype="bf16", center=False, row_offset=0, device=None, stored=False,
                 capacity=None, _norm_bound=None):
        self.engine = default_engine(device)
        self.dtype = torch_dtype(dtype)
        self.center = center
        self.row_offset = int(row_offset)
        self.norm_bound = None
        if stored:
            d = descriptors.to(self.engine.device)
            if d.dtype != self.dtype:
                raise ValueError("stored descriptors have dtype %s, expected %s" % (d.dtype, self.dtype))
            rows = d
            if _norm_bound is not None:                      # prefix(): the parent's verdict holds for its first rows
                self.norm_bound = _norm_bound[0]
            elif d.shape[0] > 0 and not self.engine.unit_rows(d):
                # starts at 1.005, not 0: rows append() normalises later are covered by the same number
                self.norm_bound = self.engine.max_row_norm(
                    d, out=torch.full((1,), 1.005, dtype=torch.float32, device=self.engine.device))
        else:
            rows = self.engine.normalize(self._as_float(descriptors), self.dtype, center)
        self._n = rows.shape[0]
        if capacity is not None and capacity > self._n:
            self._store = torch.empty((int(capacity), rows.shape[1]), dtype=self.dtype, device=self.engine.device)
            self._store[:self._n] = rows
        else:
            self._store = rows

    @classmethod
    def empty(cls, dim, capacity=4096, dtype="bf16", center=False, row_offset=0, device=None):
        """A database of `dim`-wide descriptors with no key-frames yet and room for `capacity`."""
        if dim < 1 or capacity < 1:
            raise ValueError("KeyframeDatabase.empty: dim and capacity must be positive")
        eng = default_engine(device)
        dt = torch_dtype(dtype)
        if dt not in (torch.bfloat16, torch.float16):
            raise ValueError("stored descriptor dtype must be bf16 or fp16")
        none = torch.empty((0, eng.stored_width(dim)), dtype=dt, device=eng.device)
        return cls(none, dtype=dt, center=center, row_offset=row_offset, device=device, stored=True, capacity=capacity)

    def _as_float(self, descriptors):
        x = self.engine.to_device(descriptors)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        return x

    @property
    def rows(self):
        return self._store[:self._n]

    @property
    def capacity(self):
        return self._store.shape[0]

    def __len__(self):
        return self._n

    def reserve(self, capacity):
        """Room for at least `capacity` key-frames (one device-to-device copy when it grows)."""
        if capacity > self.capacity:
            grown = torch.empty((int(capacity), self._store.shape[1]), dtype=self.dtype, device=self.engine.device)
            grown[:self._n] = self._store[:self._n]
            self._store = grown

    def append(self, descriptors):
        """Normalise and store further key-frames [B, dim]; returns their global ids (first, last+1).
        Stream-ordered on the current stream, like every other call."""
        x = self._as_float(descriptors)
        if x.dim() != 2 or self.engine.stored_width(x.shape[1]) != self._store.shape[1]:
            raise ValueError("append: descriptors must be [B, dim] with the database's width")
        b = x.shape[0]
        if self._n + b > self.capacity:
            self.reserve(max(2 * self.capacity, self._n + b))
        self.engine.normalize(x, self.dtype, self.center, out=self._store[self._n:self._n + b])
        first = self.row_offset + self._n
        self._n += b
        return first, first + b

    def prefix(self, n):
        """The first n key-frames as a database that shares this one's storage (no copy)."""
        if not 0 <= n <= self._n:
            raise ValueError("prefix: n=%d outside 0..%d" % (n, self._n))
        return KeyframeDatabase(self._store[:n], dtype=self.dtype, center=self.center, row_offset=self.row_offset,
                                device=self.engine.device, stored=True, _norm_bound=(self.norm_bound,))

    # ---- on-disk format: one .npz per shard ---------------------------------
    def save(self, path):
        """Stored rows as raw 16-bit words + what is needed to query them again."""
        np.savez(path, rows_u16=self.rows.view(torch.int16).cpu().numpy().view(np.uint16),
                 dtype=np.array(str(self.dtype).replace("torch.", "")), center=np.array(bool(self.center)),
                 row_offset=np.array(self.row_offset, dtype=np.int64), format=np.array("dlc-keyframes-v1"))

    @classmethod
    def load(cls, path, device=None, row_offset=None):
        z = np.load(path)
        if str(z["format"]) != "dlc-keyframes-v1":
            raise ValueError("%s is not a dlc-keyframes-v1 file" % path)
        dt = torch_dtype(str(z["dtype"]))
        rows = torch.from_numpy(z["rows_u16"].view(np.int16).copy()).view(dt)
        return cls(rows, dtype=dt, center=bool(z["center"]), device=device, stored=True,
                   row_offset=int(z["row_offset"]) if row_offset is None else row_offset)

    def prepare_queries(self, queries):
        x = self.engine.to_device(queries)
        if x.dtype in (torch.bfloat16, torch.float16):
            return x
        return self.engine.normalize(self._as_float(x), self.dtype, self.center)

    def tau_scale(self, q, out=None, stream=None):
        """What the certificate of stored queries q against this database multiplies tau by ([Q] device floats), or None
        when q and every row are the normaliser's (norm <= 1.005: the static tau is the bound)."""
        if self.norm_bound is None and self.engine.unit_rows(q):
            return None
        return self.engine.cosine_tau_scale(q, self.norm_bound, out=out, stream=stream)

    def match_topk(self, queries, k, out=None, details=False):
        q = self.prepare_queries(queries)
        return self.engine.match_topk(q, self.rows, k, self.row_offset, out=out, details=details, tau_scale=self.tau_scale(q))

    def match(self, queries):
        return self.engine.cosine_scores(self.prepare_queries(queries), self.rows)


class MatchPipeline:
    """Two-stream pipelined top-k match against one resident shard.  Meant for the sharded case: on a
    single GPU the score GEMM runs at the board's power cap and an overlapped selection slows it by
    more than it hides (measured -3 %), so plain KeyframeDatabase.match_topk calls are the faster form there.

    The score GEMM of batch i+1 runs on the submitting stream while the selection /
    re-score / final top-k of batch i (and, when sharded, the RCCL all-gathers of the
    per-shard results and the merge) run on a second stream: the selection is a
    latency-bound gather whose small-footprint kernel shares the CUs with the GEMM, and
    the collectives' latency leaves the critical path.  Sharded (the protocol of include/dlc.h):
      1. every rank selects its kg best groups per query and all-gathers their maxima + the best
         maximum it leaves behind ([Q, kg+1] floats);
      2. it re-scores in fp64 only the groups that can be among the kg best of the WHOLE database
         (~kg/world per query) and learns the bound B of everything all ranks left behind;
      3. the packed per-shard parts ([Q,k] int64 rows + fp64 scores, one collective) are all-gathered
         and merged in fp64 order; the merge certifies each query (k-th score > B + tau);
      4. if any query is not certified -- the same queries on every rank, the inputs of the merge
         are identical -- result() runs the exhaustive pass on every shard for those queries and
         one more all-gather + merge.  The flag reaches the host through pinned memory behind the
         batch's `done` event: no extra synchronisation on the common path.
    `depth` batches are in flight, each with its own workspace and output buffers; submit() returns
    a ticket, result(ticket) waits for that batch only.  A result must be fetched before `depth`
    further batches are submitted (its buffers are then reused), and all ranks must call submit() /
    result() in the same order (they carry collectives).  A batch whose ticket is never fetched is DROPPED when its
    slot comes round again and counted in `dropped_batches` -- unless its merge did not certify it (the exhaustive round
    of such a batch runs only inside result()): submit() then raises RuntimeError on every rank rather than overwrite an
    unverified list; (scores, idx) of result() alias the slot's buffers until then.
    """

    def __init__(self, db, k, depth=2, group=None, queries_per_batch=None, force_collectives=False):
        import torch.distributed as dist
        self.db, self.k, self.depth = db, int(k), int(depth)
        self.engine = db.engine
        dev = self.engine.device
        # the engine's side stream, made when the engine was: a stream created here, behind whatever streams the application
        # has made by now, may share a hardware queue with the submitting stream and overlap nothing (engine.py)
        self.s_select = self.engine.side_stream
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collectives: run the SHARDED protocol -- both all-gathers, the norm all-reduce, the certifying merge --
        # whatever the group's size.  A one-rank "nccl" group then takes every collective through librccl on the second
        # stream beside the score pass: the library's bring-up and stream ordering rehearsed on a single GPU
        # (python -m deeploopcloser_amd.dist --world1-smoke).  Needs an initialised process group.
        if force_collectives and not dist.is_initialized():
            raise ValueError("MatchPipeline(force_collectives=True) needs an initialised torch.distributed process group")
        self._force_collectives = bool(force_collectives)
        self._slots = []
        self._count = 0
        self._nq = queries_per_batch
        # the collective of the exchange (all_gather_into_tensor(out, inp, group=...)): torch.distributed's unless a caller
        # installs another -- bench.py's one-GPU emulation of a rank's step puts device copies here
        self.all_gather = dist.all_gather_into_tensor
        # sharded: every rank certifies with the norm bound of the WHOLE database (ranks whose rows are the normaliser's
        # contribute 1.005) -- one 4-byte all-reduce here, never read by the host
        self._norm_bound = db.norm_bound
        if self.sharded:
            r = db.norm_bound.clone() if db.norm_bound is not None else \
                torch.full((1,), 1.005, dtype=torch.float32, device=dev)
            dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
            self._norm_bound = r
        self.resolved_batches = 0          # batches that needed the exhaustive round (sharded)
        self.dropped_batches = 0           # batches whose slot was reused before result() fetched them
        self.time_collectives = False      # record events around the two all-gathers of every batch (collective_us())
        self._coll_events = []

    @property
    def sharded(self):
        """The sharded protocol (group maxima -> all-gather -> filtered re-score -> all-gather -> certifying merge) runs."""
        return self.world > 1 or self._force_collectives

    def _slot(self, i, nq, d):
        while len(self._slots) <= i:
            self._slots.append(None)
        s = self._slots[i]
        if s is None or s["nq"] != nq or s["n"] != len(self.db):       # the database may have grown (append)
            if s is not None and s["busy"]:
                s["done"].synchronize()             # its buffers are still in use on the second stream
            dev = self.engine.device
            eng, k = self.engine, self.k
            need = eng.topk_workspace_bytes(nq, len(self.db), d, k)
            s = {"nq": nq, "n": len(self.db), "ws": torch.empty(need, dtype=torch.uint8, device=dev),
                 "scores": torch.empty((nq, k), dtype=torch.float32, device=dev),
                 "idx": torch.empty((nq, k), dtype=torch.int64, device=dev),
                 "scored": torch.cuda.Event(), "done": torch.cuda.Event(), "busy": False, "fetched": True, "q": None,
                 "rows": None, "ts_buf": torch.empty((nq,), dtype=torch.float32, device=dev), "ts": None}
            if self.sharded:
                kg = eng.groups_per_query(k)
                nb = nq * k * 16                           # packed part: int64 rows [nq,k] | float64 scores [nq,k]
                s["pack"] = torch.empty(nb, dtype=torch.uint8, device=dev)
                s["p_idx"] = s["pack"][:nq * k * 8].view(torch.int64).view(nq, k)
                s["p_s64"] = s["pack"][nq * k * 8:].view(torch.float64).view(nq, k)
                s["g_pack"] = torch.empty((self.world, nb), dtype=torch.uint8, device=dev)
                s["grp_ids"] = torch.empty((nq, kg), dtype=torch.int32, device=dev)
                s["grp_max"] = torch.empty((nq, kg + 1), dtype=torch.float32, device=dev)
                s["g_max"] = torch.empty((self.world, nq, kg + 1), dtype=torch.float32, device=dev)
                s["bound"] = torch.empty((nq,), dtype=torch.float32, device=dev)
                s["m_s64"] = torch.empty((nq, k), dtype=torch.float64, device=dev)
                s["status"] = torch.empty((nq,), dtype=torch.int32, device=dev)
                s["flag"] = torch.zeros((nq,), dtype=torch.int32).pin_memory()       # the merge's per-query status, on the host
                # every shard's score pass errs by at most its plan's tau; the merge certifies with the largest tau ANY plan
                # has for this width -- the same number on every rank whatever plan its shard's size picked (a rank on the
                # bandwidth kernel's plan next to one on the MFMA plan must not certify with the smaller of the two)
                s["tau"] = eng.score_error_bound_any_plan(d)
            self._slots[i] = s
        return s

    def submit(self, queries):
        import torch.distributed as dist
        q = self.db.prepare_queries(queries)
        eng = self.engine
        i = self._count % self.depth
        s = self._slot(i, q.shape[0], q.shape[1])
        main = torch.cuda.current_stream(eng.device)
        if s["busy"]:
            if not s["fetched"]:
                # nobody asked for that batch's result.  If the merge certified it, dropping it loses nothing; if it did NOT
                # (sharded: the exhaustive round runs inside result() only), the buffers hold an unverified list that this
                # submit is about to overwrite -- say so instead of counting: the flags are identical on every rank, so
                # every rank raises here, before any collective of the new batch
                s["done"].synchronize()
                if self.sharded and bool(s["flag"].any()):
                    raise RuntimeError("MatchPipeline.submit: batch %d was never fetched and its merge did not certify %d "
                                       "quer%s -- call result(ticket) within `depth` submissions (the exhaustive round runs "
                                       "there)" % (self._count - self.depth, int((s["flag"] != 0).sum()),
                                                   "y" if int((s["flag"] != 0).sum()) == 1 else "ies"))
                self.dropped_batches += 1
            main.wait_event(s["done"])          # the workspace / outputs of this slot are free again
        s["q"] = q                              # keep the stored queries alive until the batch is done
        # ... and the database rows: append() may replace the store (reserve) while this batch's
        # selection / re-score still gathers rows from the old one on the second stream
        rows = s["rows"] = self.db.rows
        eng.score_groups(q, rows, self.k, s["ws"], stream=main)
        s["scored"].record(main)
        self.s_select.wait_event(s["scored"])
        with torch.cuda.stream(self.s_select):
            # operands that are not the normaliser's: tau follows |q| * (the database's largest row norm)
            if not self.sharded and self._norm_bound is None and eng.unit_rows(q):
                ts = s["ts"] = None
            else:
                ts = s["ts"] = eng.cosine_tau_scale(q, self._norm_bound, out=s["ts_buf"], stream=self.s_select)
            if not self.sharded:
                eng.select_topk(q, rows, self.k, s["ws"], s["scores"], s["idx"],
                                row_offset=self.db.row_offset, coop=True, stream=self.s_select, tau_scale=ts)
            else:
                eng.select_groups(q, rows, self.k, s["ws"], s["grp_ids"], s["grp_max"], coop=True,
                                  stream=self.s_select)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.time_collectives else None
                if ev:
                    ev[0].record(self.s_select)
                self.all_gather(s["g_max"].view(-1, s["g_max"].shape[-1]), s["grp_max"], group=self.group)
                if ev:
                    ev[1].record(self.s_select)
                eng.rescore_topk(q, rows, self.k, s["grp_ids"], s["grp_max"], s["p_s64"], s["p_idx"], bound=s["bound"],
                                 all_max=s["g_max"], row_offset=self.db.row_offset, coop=True, stream=self.s_select,
                                 tau_scale=ts)
                if ev:
                    ev[2].record(self.s_select)
                self.all_gather(s["g_pack"].view(-1), s["pack"], group=self.group)
                if ev:
                    ev[3].record(self.s_select)
                    self._coll_events.append(ev)
                eng.topk_merge_packed(s["g_pack"], q.shape[0], self.k, out=(s["scores"], s["idx"]), bound=s["bound"],
                                      tau=s["tau"], scores_f64=s["m_s64"], status=s["status"], tau_scale=ts)
                s["flag"].copy_(s["status"], non_blocking=True)
            s["done"].record(self.s_select)
        s["busy"] = True
        s["fetched"] = False
        self._count += 1
        return self._count - 1

    def _resolve(self, s):
        """The exhaustive round for the queries the merge could not certify (collective: every rank sees the same flags)."""
        import torch.distributed as dist
        eng, q, k = self.engine, s["q"], self.k
        with torch.cuda.stream(self.s_select):
            lower = s["m_s64"][:, k - 1].contiguous()
            eng.exhaustive_topk(q, s["rows"], k, s["ws"], lower, s["tau"], s["status"], s["p_s64"], s["p_idx"],
                                row_offset=self.db.row_offset, stream=self.s_select, tau_scale=s["ts"])
            self.all_gather(s["g_pack"].view(-1), s["pack"], group=self.group)
            eng.topk_merge_packed(s["g_pack"], q.shape[0], k, out=(s["scores"], s["idx"]), scores_f64=s["m_s64"])
            s["done"].record(self.s_select)
        s["done"].synchronize()
        self.resolved_batches += 1

    def result(self, ticket):
        if not self._count - self.depth <= ticket < self._count:
            raise ValueError("MatchPipeline.result: ticket %d is not in flight (last submitted: %d, depth %d)"
                             % (ticket, self._count - 1, self.depth))
        s = self._slots[ticket % self.depth]
        s["done"].synchronize()
        s["fetched"] = True
        if self.sharded and bool(s["flag"].any()):
            s["flag"].zero_()
            self._resolve(s)
        return s["scores"], s["idx"]

    def drain(self):
        self.s_select.synchronize()

    def collective_us(self):
        """Mean device time (microseconds, stream events around them) of the two all-gathers over the batches
        submitted while time_collectives was on: {"group_maxima": ..., "packed_topk": ..., "batches": n}; a gloo group
        stages through the host, so its figures are host round trips."""
        self.s_select.synchronize()
        ev, self._coll_events = self._coll_events, []
        if not ev:
            return None
        return {"group_maxima": float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e3,
                "packed_topk": float(np.mean([e[2].elapsed_time(e[3]) for e in ev])) * 1e3, "batches": len(ev)}


def match(desc_q, desc_db, dtype="bf16", center=False):
    """Dense cosine-similarity matrix [Q, N] (float32 numpy)."""
    db = KeyframeDatabase(desc_db, dtype=dtype, center=center)
    return db.match(desc_q).cpu().numpy()


def match_topk(desc_q, desc_db, k, dtype="bf16", center=False):
    """(scores [Q,k] float32, idx [Q,k] int64) numpy arrays."""
    db = KeyframeDatabase(desc_db, dtype=dtype, center=center)
    s, i = db.match_topk(desc_q, k)
    return s.cpu().numpy(), i.cpu().numpy()
