"""Host-side cost of the SHARDED MatchPipeline.submit (GPU box only, one GPU): world = 8 is emulated by
replacing the two all-gathers with device copies of the own part into every slot (about what an
RCCL enqueue costs the host), on one rank's 125k-row share of the 1M database."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import deeploopcloser_amd as dlc

WORLD = 8
def fake_all_gather(out, inp, group=None):
    out.view(WORLD, -1).copy_(inp.reshape(1, -1).expand(WORLD, -1))
dist.all_gather_into_tensor = fake_all_gather

eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(1)
for n in (125_000, 12_500):
    rows = eng.normalize(torch.rand((n, 4096), generator=g, device=eng.device), "bf16", center=True)
    q = eng.normalize(torch.rand((256, 4096), generator=g, device=eng.device), "bf16", center=True)
    db = dlc.KeyframeDatabase(rows, stored=True)
    pipe = dlc.MatchPipeline(db, 20, depth=3)
    pipe.world = WORLD
    for _ in range(10): pipe.submit(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): pipe.submit(q)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("rows/rank=%d world=%d (emulated): host submit %.1f us/step, end-to-end %.1f us/step" %
          (n, WORLD, t_host / 300 * 1e6, t_all / 300 * 1e6), flush=True)
    del rows, db, pipe
