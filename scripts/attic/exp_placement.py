"""Where and when the workgroups of one small LDS-DMA GEMM launch ran (GPU box only; needs
`scripts/exp_build.sh place -DDLC_EXP_DMA_PLACEMENT -DDLC_DMA_MIN_TILES=8`)."""
import ctypes as C, os, sys, collections
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
K = 2304
for M, N in ((16640, 384),):
    a = torch.rand((M, K), generator=g, device=eng.device, dtype=torch.float64)
    b = torch.rand((K, N), generator=g, device=eng.device, dtype=torch.float64)
    for _ in range(3):
        eng.gemm_bias_act(a, b, None)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 4), dtype=np.uint64)
    nwg = int(os.environ.get("DLC_NWG", "8192"))
    assert raw.dlc_exp_read_placement(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
    real = np.nonzero(buf[:, 3] == 1)[0]
    hw = buf[real, 0] & 0xffffffff
    xcc = (buf[real, 0] >> 32) & 0xf
    cu, sh, se = hw >> 8 & 0xf, hw >> 12 & 1, hw >> 13 & 0x7
    t0 = buf[real, 1].min()
    start, end = (buf[real, 1] - t0) / 100.0, (buf[real, 2] - t0) / 100.0     # 100 MHz -> us
    print("M %d N %d: %d working workgroups, launch span %.0f us" % (M, N, len(real), end.max()))
    per = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print("  distinct CUs used %d; CUs that ran 2+ workgroups %d; per-XCD workgroups %s" %
          (len(per), sum(1 for v in per.values() if v > 1), dict(sorted(collections.Counter(xcc.tolist()).items()))))
    print("  workgroup id %% 8 == XCC id for %d of %d" % (int(((real % 8) == xcc).sum()), len(real)))
    l = real // 8
    print("  SE id by (workgroup id / 8) %% 4: %s" % {k: dict(collections.Counter(se[(l % 4) == k].tolist())) for k in range(4)})
    print("  CUs per (XCC, SE): %s" % dict(sorted(collections.Counter((x, s_) for (x, s_, h, c) in per).items())))
    late = real[start > 50]
    print("  workgroups that started later than 50 us: %d  (ids %s ...)" % (len(late), late[:12].tolist()))
