"""The N>1 path on CPU: world_size-2 gloo ranks, row-sharded database, one
all-gather of per-shard top-k, k-way merge -- against the single-process oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n, d, nq, k, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeploopcloser_amd.dist import ShardedKeyframeDatabase, shard_bounds
    from oracle import cosine as ocos
    rng = np.random.RandomState(5)
    db = ocos.l2_normalize(rng.standard_normal((n, d)))
    db[n // 2 + 1] = db[1]                                  # a cross-shard exact tie
    q = ocos.l2_normalize(db[rng.choice(n, nq)] + 0.1 * rng.standard_normal((nq, d)))
    lo, hi = shard_bounds(n, world, rank)

    def local_topk(queries, kk):                           # CPU stand-in for the HIP shard match
        s, i = ocos.cosine_topk(queries.numpy(), db[lo:hi], kk, row_offset=lo)
        pad = kk - s.shape[1]
        if pad > 0:
            s = np.concatenate([s, np.full((nq, pad), -np.inf)], 1)
            i = np.concatenate([i, np.full((nq, pad), -1, dtype=np.int64)], 1)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i)

    sh = ShardedKeyframeDatabase(local_topk)
    s, i = sh.match_topk(torch.from_numpy(q), k)
    if rank == 0:
        es, ei = ocos.cosine_topk(q, db, k)
        ret["idx_equal"] = bool(np.array_equal(i.numpy(), ei))
        ret["score_err"] = float(np.abs(s.numpy() - es).max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(501, 10), (7, 5)])
def test_sharded_match_world2(n, k):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + n % 7
    mp.spawn(_worker, args=(2, port, n, 32, 9, k, ret), nprocs=2, join=True)
    assert ret["idx_equal"] and ret["score_err"] < 1e-6


def test_shard_bounds_cover_everything():
    from deeploopcloser_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 1_000_000, 1_000_003):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_merge_topk_torch_matches_oracle():
    from deeploopcloser_amd.dist import merge_topk_torch
    from oracle import cosine as ocos
    rng = np.random.RandomState(1)
    parts, q, k = 4, 6, 5
    s = rng.standard_normal((parts, q, k)).astype(np.float32)
    s[1, :, 2] = s[0, :, 1]                                   # ties across parts
    i = rng.permutation(parts * q * k).reshape(parts, q, k).astype(np.int64)
    i[3, :, 4] = -1                                           # empty slots
    ms, mi = merge_topk_torch(torch.from_numpy(s), torch.from_numpy(i), k)
    cs = np.transpose(s, (1, 0, 2)).reshape(q, parts * k).astype(np.float64)
    ci = np.transpose(i, (1, 0, 2)).reshape(q, parts * k)
    cs = np.where(ci < 0, -np.inf, cs)
    ci2 = np.where(ci < 0, np.iinfo(np.int64).max, ci)
    es, ei = ocos.merge_topk(cs, ci2, k)
    assert np.array_equal(mi.numpy(), ei)
    assert np.allclose(ms.numpy(), es)
