R=$GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
N, P, H, sb = 1063, 30, 2500, 32
x = torch.rand((N, P, 1681), generator=g, device=eng.device, dtype=torch.float64)
desc = dlc.SDAV(seed=1).transform_tensor(x).reshape(N, P, H)
score = eng.distinctive_score(desc, 0.5, 0.2)
def plain(sb=sb):
    det = dlc.SdavLoopClosureDetector(score, patches=P, width=H, k=5, exclusion=30, capacity=N)
    outs = [det.query_and_insert(desc[lo:lo + sb]) for lo in range(0, N, sb)]
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
def piped(sb=sb):
    det = dlc.SdavLoopClosureDetector(score, patches=P, width=H, k=5, exclusion=30, capacity=N)
    outs, prev = [], None
    for lo in range(0, N, sb):
        t = det.submit(desc[lo:lo + sb])
        if prev is not None: outs.append(det.result(prev))
        prev = t
    outs.append(det.result(prev))
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts, hs = [], []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); hs.append((time.perf_counter() - t0) * 1e3); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("   host enqueue %.2f ms of %.2f" % (min(hs), min(ts)))
    return min(ts), sorted(ts)[len(ts)//2], r
for b in (32, 64, 16):
    a0, a1, ra = timed(lambda: plain(b)); b0, b1, rb = timed(lambda: piped(b))
    same = torch.equal(ra[1], rb[1]) and torch.equal(torch.nan_to_num(ra[0], posinf=1e300, neginf=-1e300), torch.nan_to_num(rb[0], posinf=1e300, neginf=-1e300))
    print("batches of %d: batch by batch %.2f ms (median %.2f), two in flight %.2f ms (median %.2f), same lists %s" % (b, a0, a1, b0, b1, same), flush=True)
PY
