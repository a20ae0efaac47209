"""What ONE rank of an N-GPU run does per 256-query batch, on one GPU (GPU box only): MatchPipeline's sharded protocol
over a shard of 1 M / N rows with the two all-gathers replaced by device copies that fabricate the other ranks'
contributions (their group maxima = this rank's, jittered; their packed top-k = this rank's), so that the filter keeps
about kg / N groups as it does with real shards.  No RCCL, no other GPU: the number is the GPU-side floor of a rank's
step, i.e. an upper bound on what N GPUs can reach (256 / step time x 1, whole job = the same, every rank in lockstep).
Usage: python scripts/exp_shard_step.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import deeploopcloser_amd as dlc

eng = dlc.default_engine(0)
g = torch.Generator(device=eng.device); g.manual_seed(0)
d, nq, k, total = 4096, 256, 20, 1_000_000


def fake_all_gather(out, inp, group=None):
    parts = out.numel() // inp.numel()
    o = out.view(parts, -1)
    o.copy_(inp.reshape(1, -1).expand(parts, -1))
    if inp.dtype == torch.float32 and out.shape[-1] != inp.numel():    # the group maxima: other ranks' differ a little
        o[1:] *= 1.0 + 0.02 * (torch.rand(o[1:].shape, generator=g, device=o.device) - 0.5)


dist.all_gather_into_tensor = fake_all_gather
base = None
for parts in (1, 2, 4, 8):
    rows = total // parts
    db = dlc.KeyframeDatabase(torch.randn((rows, d), generator=g, device=eng.device), dtype="bf16")
    q = eng.normalize(torch.randn((nq, d), generator=g, device=eng.device), "bf16")
    pipe = dlc.MatchPipeline(db, k, depth=3 if parts > 1 else 2)
    pipe.world = parts                                                   # the sharded branch of submit()
    for _ in range(30):
        t = pipe.submit(q)
    pipe.result(t); torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    for _ in range(steps):
        t = pipe.submit(q)
    pipe.result(t); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if parts == 1:
        base = ms
    print("N = %d: %7d rows per rank, %.3f ms per 256-query batch per rank -> at most %.0f k query-frames/s for the job "
          "(%.2fx of one GPU's pipelined %.3f ms; ideal %dx)" % (parts, rows, ms, nq / ms, base / ms, base, parts), flush=True)
    del db, pipe
    torch.cuda.empty_cache()
