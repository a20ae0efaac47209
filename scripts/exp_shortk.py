"""LDS-DMA kernel vs register-staged kernel (forced by an odd row stride of A) on GEMMs with a short K, e.g. the weight
gradients of an SDAV training step (1681 x 2500 x 300) (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L
eng = dlc.default_engine()
g = torch.Generator(device=eng.device); g.manual_seed(0)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


for (m, n, k) in ((1681, 2500, 300), (2500, 2500, 300), (1682, 2500, 300), (1682, 2500, 600), (1682, 2500, 1200), (1682, 2500, 2500),
                  (900, 2500, 2500), (3000, 2500, 2500), (6000, 2500, 2500), (3000, 2500, 300), (6000, 2500, 300)):
    a = torch.rand((m, k), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((k, n), generator=g, device=eng.device, dtype=torch.float64)
    out = torch.empty((m, n), dtype=torch.float64, device=eng.device)
    wide = torch.zeros((m, k + 1), dtype=torch.float64, device=eng.device)
    wide[:, :k] = a
    call = lambda A, lda: eng._check(eng.lib.dlc_gemm_bias_act(eng.ctx, L.DLC_F64, L.DLC_B_KN, 0, m, n, k, A.data_ptr(), lda,
                                                               b.data_ptr(), n, None, out.data_ptr(), n, None))
    t_dma = timed(lambda: call(a, k))
    t_old = timed(lambda: call(wide, k + 1))
    print("M %5d N %5d K %5d  tiles256 %4d  default %.1f us   register-staged %.1f us" %
          (m, n, k, -(-m // 256) * -(-n // 128), t_dma, t_old), flush=True)
