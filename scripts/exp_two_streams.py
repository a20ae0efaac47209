"""Do two half batches on two streams beat one whole batch (the halves' partial rounds filling each other) (GPU box only)?
SDAV.transform and CnnVtl.transform of 1063 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)
N = 1063
side = torch.cuda.Stream(device=eng.device)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def split(fn, x, parts):
    main = torch.cuda.current_stream()
    outs = [None] * parts
    step = -(-x.shape[0] // parts)
    side.wait_stream(main)
    for p in range(parts):
        xs = x[p * step:(p + 1) * step]
        if p % 2:
            with torch.cuda.stream(side):
                outs[p] = fn(xs)
        else:
            outs[p] = fn(xs)
    main.wait_stream(side)
    return outs


x = torch.rand((N, 30, 1681), generator=g, device=eng.device, dtype=torch.float64)
net = dlc.SDAV(seed=1)
print("SDAV.transform   whole %.2f ms   2 streams x 2 parts %.2f ms   x 4 parts %.2f ms" %
      (timed(lambda: net.transform_tensor(x)), timed(lambda: split(net.transform_tensor, x, 2)), timed(lambda: split(net.transform_tensor, x, 4))), flush=True)
del x
fr = torch.randint(0, 256, (N, 192, 240, 3), generator=g, device=eng.device).to(torch.float64)
cnn = dlc.CnnVtl(input_shape=[N, 192, 240, 3])
print("CnnVtl.transform whole %.2f ms   2 streams x 2 parts %.2f ms   x 4 parts %.2f ms" %
      (timed(lambda: cnn.transform_tensor(fr)), timed(lambda: split(cnn.transform_tensor, fr, 2)), timed(lambda: split(cnn.transform_tensor, fr, 4))), flush=True)
