"""Where the waves of gemm_dma_f64_kernel spend their cycles (GPU box only; needs the diagnostic build
`scripts/exp_build.sh stamps -DDLC_EXP_DMA_STAMPS`): per K tile, cycles waiting for the tile's DMA (s_waitcnt vmcnt),
in the barrier, and in the tile body (LDS reads + 64 MFMAs + DMA issue), for waves 0-3 and their partners 4-7."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeploopcloser_amd as dlc
from deeploopcloser_amd import _lib as L

L._lib = None
L.LIB_PATH = os.path.abspath(sys.argv[1])
eng = dlc.default_engine(0)
raw = C.CDLL(L.LIB_PATH)
g = torch.Generator(device=eng.device); g.manual_seed(0)
for name, m, n, k in (("SDAV layer", 31890, 2500, 2500), ("conv3-like plain", 65520, 384, 2304)):
    a = torch.rand((m, k), generator=g, device=eng.device, dtype=torch.float64)
    w = torch.randn((k, n), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.zeros((n,), device=eng.device, dtype=torch.float64)
    for _ in range(3):
        eng.gemm_bias_act(a, w, b, act=1)
    torch.cuda.synchronize()
    buf = np.zeros((256, 8, 4), dtype=np.uint64)
    assert raw.dlc_exp_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
    ok = buf[:, 0, 3] > 0
    s = buf[ok].astype(np.float64)
    per = s[:, :, :3] / s[:, :, 3:4]
    print(name, "K tiles", int(s[0, 0, 3]) - 1, "workgroups sampled", int(ok.sum()))
    for grp, sl in (("waves 0-3 (early)", slice(0, 4)), ("waves 4-7 (late) ", slice(4, 8))):
        w_, b_, body = per[:, sl, 0].mean(), per[:, sl, 1].mean(), per[:, sl, 2].mean()
        print("  %s  wait for DMA %6.0f  barrier %6.0f  tile body %6.0f  total %6.0f cycles per K tile (ideal 8192 = 2 waves x 64 MFMAs x 64)"
              % (grp, w_, b_, body, w_ + b_ + body))
