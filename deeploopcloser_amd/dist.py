"""Multi-GPU match: the key-frame database is row-sharded over the ranks of one
node (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI),
queries are replicated, every rank matches against its shard with global row
offsets, and the per-shard parts are all-gathered and merged with the same
ordering rule (fp64 score key, then the lower global row).  On GPUs there is ONE
implementation of the exchange -- matching.MatchPipeline (group maxima first,
then the packed parts, a certifying merge): ShardedKeyframeDatabase.from_database
delegates to it.  The shard arithmetic and the merge rule are also exercised on
CPU with the gloo backend (tests/test_dist_cpu.py) through a pluggable
`local_topk`.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous row range [lo, hi) of `rank`: the first n % world ranks get one extra row."""
    base, extra = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_topk_torch(scores, idx, k):
    """Reference merge on any device: [parts, Q, k] -> [Q, k], score desc, ties -> lower index
    (empty slots carry idx < 0)."""
    parts, q, kk = scores.shape
    s = scores.permute(1, 0, 2).reshape(q, parts * kk).to(torch.float64)
    i = idx.permute(1, 0, 2).reshape(q, parts * kk)
    s = torch.where(i < 0, torch.full_like(s, float("-inf")), s)
    big = torch.iinfo(torch.int64).max
    i_key = torch.where(i < 0, torch.full_like(i, big), i)
    # lexicographic sort: first by index, then stable by score descending
    o1 = torch.argsort(i_key, dim=1, stable=True)
    s1, i1 = torch.gather(s, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(-s1, dim=1, stable=True)
    s2, i2 = torch.gather(s1, 1, o2)[:, :k], torch.gather(i1, 1, o2)[:, :k]
    return s2.to(scores.dtype), i2


class ShardedKeyframeDatabase:
    """Rank-local shard + the exchange.

    from_database(db): the GPU form -- match_topk() is one MatchPipeline batch (submit + result).
    ShardedKeyframeDatabase(local_topk): local_topk(queries, k) -> (scores [Q,k], idx [Q,k] i64 GLOBAL)
    runs the shard-local match (scores in the precision the order is to be decided in); one all-gather
    each of scores and rows and merge_topk_torch follow -- the CPU / gloo form.
    """

    def __init__(self, local_topk, merge=None, group=None):
        self.local_topk = local_topk
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gather_s = None
        self._gather_i = None
        self._db = None
        self._pipes = {}

    @classmethod
    def from_database(cls, db, group=None):
        obj = cls(None, group=group)
        obj._db = db
        return obj

    def match_topk(self, queries, k):
        if self._db is not None:
            if self.world == 1:                      # nothing to exchange: the one-shot call (faster than the pipeline on one GPU)
                return self._db.match_topk(queries, k)
            from .matching import MatchPipeline
            pipe = self._pipes.get(k)
            if pipe is None:
                pipe = self._pipes[k] = MatchPipeline(self._db, k, depth=1, group=self.group)
            # (the pipeline hands out its slot's reusable buffers; every branch of this method returns tensors the
            # caller owns, so the next call must not overwrite them)
            s, i = pipe.result(pipe.submit(queries))
            return s.clone(), i.clone()
        s, i = self.local_topk(queries, k)
        if self.world == 1:
            return s, i
        if self._gather_s is None or self._gather_s.shape != (self.world,) + tuple(s.shape):
            self._gather_s = torch.empty((self.world,) + tuple(s.shape), dtype=s.dtype, device=s.device)
            self._gather_i = torch.empty((self.world,) + tuple(i.shape), dtype=i.dtype, device=i.device)
        # concatenation form (works on RCCL and gloo alike): [world*Q, k] viewed as [world, Q, k]
        dist.all_gather_into_tensor(self._gather_s.view(-1, s.shape[-1]), s.contiguous(), group=self.group)
        dist.all_gather_into_tensor(self._gather_i.view(-1, i.shape[-1]), i.contiguous(), group=self.group)
        if self.merge is not None:
            return self.merge(self._gather_s, self._gather_i, k)
        return merge_topk_torch(self._gather_s, self._gather_i, k)


def world1_smoke(rows=131072, dim=4096, queries=256, k=20, backend="nccl", batches=6, device=0):
    """RCCL first contact on ONE GPU: a one-rank process group on `backend` ("nccl" is RCCL on ROCm), then the sharded
    protocol with every collective forced through the library (MatchPipeline(force_collectives=True): the norm all-reduce,
    the all-gather of the group maxima and the all-gather of the packed parts, on the second stream beside the score
    pass) -- each batch must equal the one-shot call on the same operands bit for bit; then the same with a crowded
    database (a tie no merge can certify), which takes the exhaustive round and its third all-gather.  Returns a dict
    (what bench.py puts in its line as `rccl_world1_smoke`).  Run it in a process of its own:
    `python -m deeploopcloser_amd.dist --world1-smoke` prints the dict as one JSON line."""
    import datetime
    import os
    import socket
    import time
    from .matching import KeyframeDatabase, MatchPipeline
    from .engine import default_engine

    torch.cuda.set_device(device)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    t0 = time.perf_counter()
    kw = {"device_id": torch.device("cuda", device)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=0, world_size=1, timeout=datetime.timedelta(seconds=60), **kw)
    t_init = time.perf_counter() - t0
    try:
        eng = default_engine(device)
        t0 = time.perf_counter()
        one = torch.ones(1, dtype=torch.int32, device=eng.device)
        dist.all_reduce(one)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        if int(one.item()) != 1:
            raise RuntimeError("all-reduce over one rank gave %d" % int(one.item()))
        g = torch.Generator(device=eng.device)
        g.manual_seed(7)
        stored = torch.empty((rows, eng.stored_width(dim)), dtype=torch.bfloat16, device=eng.device)
        for lo in range(0, rows, 32768):
            hi = min(rows, lo + 32768)
            eng.normalize(torch.rand((hi - lo, dim), generator=g, device=eng.device), "bf16", center=True, out=stored[lo:hi])
        pick = torch.randint(0, rows, (queries,), generator=g, device=eng.device)
        # queries: stored rows moved a little (unit rows: a component is ~ 1 / sqrt(dim)) -- their neighbour stays the best row
        near = stored[pick][:, :dim].float()
        q = eng.normalize(near + (0.3 / dim ** 0.5) * torch.randn((queries, dim), generator=g, device=eng.device), "bf16")
        out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": t_init, "first_collective_s": t_first,
               "db_rows": rows, "dim": dim, "queries": queries, "k": k}
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out["rccl_version"] = None

        def check(db, label):
            want = db.match_topk(q, k, details=True)
            pipe = MatchPipeline(db, k, depth=2, force_collectives=True)
            pipe.time_collectives = True
            same = True
            for b in range(batches):
                t = pipe.submit(q)
                if t >= 1:
                    s, i = pipe.result(t - 1)
                    same = same and bool(torch.equal(i, want.idx) and torch.equal(s, want.scores))
            s, i = pipe.result(batches - 1)
            same = same and bool(torch.equal(i, want.idx) and torch.equal(s, want.scores))
            out[label + "_equals_one_shot"] = same
            out[label + "_collective_us"] = pipe.collective_us()
            out[label + "_resolved_batches"] = pipe.resolved_batches
            return same

        ok = check(KeyframeDatabase(stored, dtype="bf16", stored=True), "pipeline")
        # a tie no selection can certify: kg * 8 + 1 exact copies of query 0's row, spread over the database
        copies = eng.groups_per_query(k) * 8 + 1
        where = torch.linspace(0, rows - 1, copies, device=eng.device).long().unique()
        stored[where] = stored[pick[0]].clone()
        ok = check(KeyframeDatabase(stored, dtype="bf16", stored=True), "crowded") and ok
        ok = ok and out["crowded_resolved_batches"] == batches and out["pipeline_resolved_batches"] == 0
        out["ok"] = bool(ok)
        return out
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import argparse
    import json
    import sys
    ap = argparse.ArgumentParser(description="multi-GPU helpers: --world1-smoke = RCCL first contact on one GPU")
    ap.add_argument("--world1-smoke", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--dim", type=int, default=4096)
    a = ap.parse_args()
    if not a.world1_smoke:
        ap.print_help()
        sys.exit(2)
    res = world1_smoke(rows=a.rows, dim=a.dim, backend=a.backend)
    print(json.dumps(res), flush=True)
    sys.exit(0 if res["ok"] else 1)
