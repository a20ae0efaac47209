"""Multi-GPU match: the key-frame database is row-sharded over the ranks of one
node (one process per GPU, torch.distributed backend "nccl" = RCCL over xGMI),
queries are replicated, every rank runs the fused top-k on its shard with
global row offsets, and ONE all-gather of the per-shard [Q,k] (score, index)
pairs (Q*k*12 bytes per rank, latency-bound) is followed by a k-way merge with
the same ordering rule.  No other collective is on the data path.

The shard arithmetic and the merge are also exercised on CPU with the gloo
backend (tests/test_dist_cpu.py) using a pluggable `local_topk`.
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
    """Rank-local shard + the all-gather / merge step.

    local_topk(queries, k) -> (scores [Q,k] f32, idx [Q,k] i64 GLOBAL) runs the
    shard-local match; on GPUs it is KeyframeDatabase.match_topk, and the merge
    is the HIP dlc_topk_merge kernel.
    """

    def __init__(self, local_topk, merge=None, group=None):
        self.local_topk = local_topk
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._gather_s = None
        self._gather_i = None

    @classmethod
    def from_database(cls, db, group=None):
        eng = db.engine
        return cls(lambda q, k: db.match_topk(q, k), merge=lambda s, i, k: eng.topk_merge(s, i), group=group)

    def match_topk(self, queries, k):
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
