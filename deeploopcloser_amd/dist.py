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
