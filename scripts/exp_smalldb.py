import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
for (q, n, d) in [(1063, 1063, 75008), (256, 12500, 4096), (16, 5000, 8192), (300, 700, 4160), (1063, 1063, 4096), (64, 3000, 16384)]:
    x = torch.randn((n, d), generator=g, device=eng.device).to(torch.bfloat16)
    qs = x[:q].contiguous() if q <= n else torch.randn((q, d), generator=g, device=eng.device).to(torch.bfloat16)
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize(); b = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); b = min(b, e0.elapsed_time(e1))
        return b
    print((q, n, d), "scores %.3f ms  top20 %.3f ms" % (t(lambda: eng.cosine_scores(qs, x)), t(lambda: eng.match_topk(qs, x, 20))), flush=True)
