"""Which property of the Gram launch costs it 15 % against an SDAV layer on the same kernel (GPU box only): plain
fp64 GEMMs through dlc_gemm_bias_act of M = 31890, K = 2500 and growing N, B either a fresh [K,N] matrix, the
transposed A (the Gram's operand) or A itself as an [N,K] operand.  Library: argv[1] (default: the shipped one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

if len(sys.argv) > 1:
    L._lib = None
    L.LIB_PATH = os.path.abspath(sys.argv[1])
eng = dlc.default_engine(0)
g = torch.Generator(device=eng.device); g.manual_seed(0)
M, K = 31890, 2500
a = torch.rand((M, K), generator=g, device=eng.device, dtype=torch.float64)
at = a.t().contiguous()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for n in (2500, 5000, 10000, 20000, 31890):
    out = torch.empty((M, n), device=eng.device, dtype=torch.float64)
    w = torch.randn((K, n), generator=g, device=eng.device, dtype=torch.float64)
    cases = [("fresh [K,N]", w, L.DLC_B_KN), ("A^T   [K,N]", at[:, :n], L.DLC_B_KN), ("A     [N,K]", a[:n], L.DLC_B_NK)]
    for name, b, lay in cases:
        if lay == L.DLC_B_KN and not b.is_contiguous():
            # a column slice of A^T keeps A^T's row stride: the very operand the Gram launch reads
            fn = lambda b=b, lay=lay: eng._check(eng.lib.dlc_gemm_bias_act(
                eng.ctx, L.DLC_F64, lay, L.DLC_ACT_NONE, M, n, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                None, out.data_ptr(), out.stride(0), eng._stream()))
        else:
            fn = lambda b=b, lay=lay: eng.gemm_bias_act(a, b, None, act=L.DLC_ACT_NONE, blayout=lay, out=out)
        ms = timed(fn)
        print("N %6d  %s  %.3f ms  %.1f TF" % (n, name, ms, 2.0 * M * n * K / ms / 1e9), flush=True)
    del out, w
